// oracle/ref_helper_driver.cpp -- TEST INFRASTRUCTURE ONLY.
// Thin extern "C" driver around the GENUINE reference header cpu_version/helper.hpp, compiled from
// where it lies under /root/reference (see oracle/Makefile: -I$(REF)/cpu_version).  helper.hpp has no
// Eigen dependency, so it builds with plain g++ and no stand-ins.  Output goes to oracle/_ref/ only.
// Used to pin the oracle's restatement of code_t / extractDistance / calcRatio / pow<uint>.
#include <cstdint>
#include <cstring>
#include "helper.hpp"

extern "C" {
float ref_extract_distance(float a, float b, float c, float l) { return extractDistance(a, b, c, l); }
float ref_calc_ratio(float a, float b, float c) { return calcRatio<float>(a, b, c); }
uint32_t ref_code_pack(unsigned a, unsigned b, float l) {
  code_t c((unsigned char)a, (unsigned char)b, l);
  uint32_t r; std::memcpy(&r, &c.raw, 4); return r;
}
unsigned ref_code_a(uint32_t raw) { code_t c; std::memcpy(&c.raw, &raw, 4); return c.a(); }
unsigned ref_code_b(uint32_t raw) { code_t c; std::memcpy(&c.raw, &raw, 4); return c.b(); }
float ref_code_lambda(uint32_t raw) { code_t c; std::memcpy(&c.raw, &raw, 4); return c.lambda(); }
unsigned short ref_to_ushort(float f) { code_t c; return c.toUShort(f); }
unsigned ref_upow(unsigned x, unsigned n) { return pow<uint>(x, n); }
unsigned ref_sizeof_code() { return (unsigned)sizeof(code_t); }
}
