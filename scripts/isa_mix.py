"""Static instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing.  usage: python scripts/isa_mix.py FILE.s SYMBOL_SUBSTRING [top]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if re.match(r'^\S*' + re.escape(sub) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
ins = []
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':') or re.match(r'^\S+:\s', t):
        continue
    ins.append(t.split()[0])
c = Counter(ins)
cat = Counter()
for k, v in c.items():
    cat['valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'lds' if k.startswith('ds_') else
        'vmem' if k.split('_')[0] in ('global', 'buffer', 'scratch', 'flat') else 'other'] += v
print(lines[start][:90], "\nstatic instructions:", len(ins), dict(cat))
print(", ".join(f"{k} {v}" for k, v in c.most_common(top)))
