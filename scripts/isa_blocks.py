"""Per-basic-block instruction mix of one kernel in a hipcc -S listing: python scripts/isa_blocks.py file.s SYMBOL_SUBSTRING"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M)
body = m.group(2).split('\n')
blocks, cur, name = [], [], 'entry'
for l in body:
    t = l.strip()
    if re.match(r'^\.LBB\S+:', t):
        blocks.append((name, cur)); name = t.split(':')[0]; cur = []
    elif t and not t.startswith(';') and not t.startswith('.'):
        cur.append(t)
blocks.append((name, cur))
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for n, b in blocks:
    if len(b) < minlen: continue
    c = lambda p: sum(1 for x in b if x.startswith(p))
    print(f'{n:12s} n={len(b):4d} valu={c("v_"):4d} f64add={c("v_add_f64"):3d} min={c("v_min_f64"):3d} cmp={c("v_cmp"):3d} '
          f'ds_r={c("ds_read"):3d} ds_w={c("ds_write"):3d} sstore={c("s_store"):3d} salu={c("s_") - c("s_store") - c("s_waitcnt"):3d} '
          f'wait={c("s_waitcnt"):3d} vmem={c("global_") + c("buffer_"):3d}')
if len(sys.argv) > 4:
    for n, b in blocks:
        if n == sys.argv[4]: print('\n'.join(b))
