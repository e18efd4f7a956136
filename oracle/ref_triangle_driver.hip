// oracle/ref_triangle_driver.hip -- TEST INFRASTRUCTURE ONLY.
// Thin extern "C" driver around the GENUINE reference header pqt/triangle.cuh (its __host__ side),
// compiled by hipcc from where it lies under /root/reference.  Output goes to oracle/_ref/ only.
// Pins the lambda codec / dist / project known answers of run.cu:33-113.
#include <hip/hip_runtime.h>
#include <sys/types.h>
#include <math.h>
#include "triangle.cuh"

extern "C" {
unsigned short reftri_to_ushort(float f) { return pqt::toUShort(f); }
float reftri_to_float(unsigned short s) { return pqt::toFloat(s); }
float reftri_dist(float a2, float b2, float c2, float l) { return pqt::dist(a2, b2, c2, l); }
float reftri_project(float a2, float b2, float c2) { return pqt::project(a2, b2, c2); }
float reftri_project_d2(float a2, float b2, float c2, float* d2) { volatile float d = 0; float l = pqt::project(a2, b2, c2, d); *d2 = d; return l; }
int reftri_equal(float a, float b) { return pqt::equal(a, b) ? 1 : 0; }
}
