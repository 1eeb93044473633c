"""Static instruction census of one kernel: instructions per source line (from .loc) between s_barriers.
usage: isa_lines.py <asm with -gline-tables-only> <mangled-name-prefix>"""
import re, sys, collections
asm, prefix = sys.argv[1], sys.argv[2]
files = {}
inside = False
cur = (0, 0)
seg = 0
segs = collections.defaultdict(lambda: collections.Counter())
seg_start = {0: 0}
kinds = collections.defaultdict(lambda: collections.Counter())
n = 0
for ln, line in enumerate(open(asm), 1):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', line) or re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"', line)
    if m:
        files[int(m.group(1))] = m.group(2).split('/')[-1]
        continue
    if not inside:
        if line.startswith(prefix) and ':' in line:
            inside = True
        continue
    if line.startswith('.Lfunc_end') or line.startswith('\t.section'):
        break
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    s = line.strip()
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
        continue
    op = s.split()[0]
    n += 1
    segs[seg][cur] += 1
    k = 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')) else 'valu'
    kinds[seg][k] += 1
    if op == 's_barrier':
        seg += 1
        seg_start[seg] = n
print('total instructions', n)
for sg in sorted(segs):
    c = segs[sg]
    tot = sum(c.values())
    print(f'--- segment {sg}: {tot} instrs  {dict(kinds[sg])}')
    byfile = collections.Counter()
    for (f, l), v in c.items():
        byfile[(files.get(f, f), l)] += v
    for (f, l), v in byfile.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 12):
        print(f'     {v:5d}  {f}:{l}')
