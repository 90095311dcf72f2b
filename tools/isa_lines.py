"""Static instruction count per source line of one kernel: python tools/isa_lines.py build/isa_g/k.s ILb0ELb0"""
import re, sys, collections
path, tag = sys.argv[1], sys.argv[2]
src = open("path_tracer_amd/csrc/pt_path_kernel.hip").read().splitlines()
inside = False; line = 0
cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
for l in open(path):
    s = l.strip()
    if s.startswith(".type") and "pt_path_kernel" in s and tag in s: inside = True; continue
    if inside and s.startswith(".Lfunc_end"): break
    if not inside: continue
    m = re.match(r"\.loc\s+\d+\s+(\d+)", s)
    if m: line = int(m.group(1)); continue
    if not s or s.startswith((".", ";")) or s.endswith(":"): continue
    op = s.split()[0]
    k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "mem"
    if op.startswith(("s_load", "global_", "buffer_", "ds_", "flat_")): k = "mem"
    cnt[line] += 1; kinds[line][k] += 1
tot = collections.Counter()
for ln in sorted(cnt):
    k = kinds[ln]; tot.update(k)
    print(f"{ln:4d} {cnt[ln]:4d} v{k['valu']:4d} s{k['salu']:4d} m{k['mem']:3d} | {src[ln-1].strip()[:110] if 0 < ln <= len(src) else ''}")
print("total", dict(tot))
