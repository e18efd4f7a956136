#!/bin/bash
# Round 6 (no device): static evidence for the opt-in kernels -- registers, spills, LDS and the instruction mix of each kernel body
# (hipcc -S --cuda-device-only for gfx950), old kernel beside new.  Output: profiles/r06_static_isa.txt
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/product-quantization-tree_amd/csrc; T=$(mktemp -d /tmp/isa6_XXXX)
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-result -S --cuda-device-only"
hipcc $F -o $T/sr.s $C/pqt_shared_launch.hip 2>/dev/null &
hipcc $F -o $T/tr.s $C/pqt_traverse_launch.hip 2>/dev/null &
hipcc $F -o $T/rr.s $C/pqt_rerank_launch.hip 2>/dev/null &
wait
python3 - $T > $ROOT/profiles/r06_static_isa.txt <<'PY'
import re, sys, collections
T = sys.argv[1]
def kernels(fn):
    txt = open(fn).read()
    meta = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
        blk = m.group(2)
        g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, blk) or [None, '?'])[1]
        meta[m.group(1)] = dict(vgpr=g('vgpr_count'), vspill=g('vgpr_spill_count'), sspill=g('sgpr_spill_count'), lds_static=g('group_segment_fixed_size'))
    body = {}
    cur = None
    for l in txt.split('\n'):
        m = re.match(r'^(_Z\w+):', l)
        if m: cur = m.group(1); body[cur] = collections.Counter(); continue
        if cur:
            t = l.strip()
            if t.startswith('s_endpgm'): cur = None; continue
            if not t or t.startswith((';', '.')) or t.endswith(':'): continue
            body[cur][t.split()[0]] += 1
    return meta, body
def cls(m):
    if m.startswith('ds_'): return 'LDS'
    if m.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'VMEM'
    if m.startswith('v_'): return 'VALU'
    if m.startswith('s_waitcnt'): return 'WAIT'
    if m.startswith('s_nop'): return 'NOP'
    return 'SALU'
def show(fn, pats, top=8):
    meta, body = kernels(fn)
    for k in body:
        if not any(re.search(p, k) for p in pats): continue
        c = collections.Counter()
        for m, n in body[k].items(): c[cls(m)] += n
        md = meta.get(k, {})
        print("%s\n   VGPR %s  spilled VGPR dwords %s  spilled SGPRs %s  static LDS %s B   instructions in the body: %d  %s" %
              (k, md.get('vgpr'), md.get('vspill'), md.get('sspill'), md.get('lds_static'), sum(c.values()), dict(c)))
        print("   " + "  ".join("%s x%d" % (m, n) for m, n in body[k].most_common(top)))
print("# static view of the round-6 kernels (whole kernel bodies, every unrolled copy counted once; not executed counts).  scripts/r06_static_isa.sh\n")
print("## evaluating kernel of the shared-row pass: pqt_k_sr_adc (default) vs pqt_k_sr_adc2 (sr_kernel = 2).  Both hold TWO copies of the evaluate body")
print("## (register sets A / B); sr_adc's copy serves ONE query per pass of its loop (64 ds_read_b32), sr_adc2's serves up to EIGHT (256 ds_read_b64).")
show(T + '/sr.s', [r'pqt_k_sr_adcI', r'pqt_k_sr_adc2I'])
print("\n## selection scan: pqt_k_sr_select<.., PHASE 2> (default) vs pqt_k_sr_scan_seg<16, SEG, QD> + pqt_k_sr_merge<4, SEG>")
show(T + '/sr.s', [r'pqt_k_sr_selectILi16.*Li2EEv', r'pqt_k_sr_scan_seg', r'pqt_k_sr_merge'])
print("\n## filtered rerank on shards: pqt_k_rerank_select<12, 8, 2, false, SH, 6, 2, true> (default, 12 wavefronts per CU) vs pqt_k_pair_scan (16 per CU) + band launch")
show(T + '/rr.s', [r'pqt_k_rerank_selectILi12ELi8ELi2ELb0ELb[01]ELi6ELi2ELb1'])
show(T + '/sr.s', [r'pqt_k_pair_scan', r'pqt_k_sr_selectILi12.*Li3EEv'])
print("\n## wide traversal with the LDS first level (the compacted probing, filter_l1 = 2, is a run-time branch of the same kernels)")
show(T + '/tr.s', [r'pqt_k_traverse_f1'])
PY
rm -rf $T; wc -l $ROOT/profiles/r06_static_isa.txt
