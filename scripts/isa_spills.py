"""Where a kernel spills: instruction count, scratch loads/stores by position, and the loops (backward branches) with the scratch
operations inside each -- from the assembly of a device-only compile:
    hipcc ... --cuda-device-only -S -o /tmp/k.s file.hip ; python scripts/isa_spills.py /tmp/k.s <substring of the mangled kernel name>"""
import collections, re, sys
t = open(sys.argv[1]).read()
want = sys.argv[2]
parts = re.split(r'\n(_Z\w+):[^\n]*\n', t)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
    if want not in name: continue
    lines = body.split('\n')
    isins = lambda l: l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')
    ins = [l for l in lines if isins(l)]
    pos = [(k, l.strip().split()[0]) for k, l in enumerate(ins) if 'scratch_' in l]
    print(name[:90], '\n instructions', len(ins), 'scratch ops', len(pos))
    h = collections.Counter((k * 20 // len(ins), op) for k, op in pos)
    print(' by twentieth of the code:', dict(sorted(h.items())))
    labels, k = {}, 0
    for l in lines:
        if l.startswith('.LBB') and l.rstrip().endswith(':'): labels[l.split(':')[0]] = k
        elif isins(l): k += 1
    k = 0
    for l in lines:
        if isins(l):
            m = re.search(r's_c?branch\w*\s+(\.LBB\S+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < k and k - labels[m.group(1)] > 40:
                a = labels[m.group(1)]
                print(' loop %6d -> %6d  len %5d  scratch ops inside %d' % (a, k, k - a, sum(1 for kk, _ in pos if a <= kk <= k)))
            k += 1
