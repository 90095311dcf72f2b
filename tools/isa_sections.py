"""Static instruction count of one kernel attributed to the line of the KERNEL BODY each
instruction was inlined into (llvm-symbolizer --inlines on every instruction address).
  hipcc ... -gline-tables-only --cuda-device-only --no-gpu-bundle-output -c pt_path_kernel.hip -o k2.o
  python tools/isa_sections.py build/isa_g/k2.o ILb0ELb0"""
import collections, re, subprocess, sys
obj, tag = sys.argv[1], sys.argv[2]
LLVM = "/opt/rocm/lib/llvm/bin/"
dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
inside = False; insts = []
for l in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
    if m: inside = ("pt_path_kernel" in m.group(1) and tag in m.group(1)); continue
    if not inside: continue
    m = re.match(r"^\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):", l)
    if m: insts.append((int(m.group(2), 16), m.group(1)))
addrs = "\n".join(hex(a) for a, _ in insts)
sym = subprocess.run([LLVM + "llvm-symbolizer", "--inlines", "--obj=" + obj, "--output-style=LLVM"], input=addrs, capture_output=True, text=True).stdout
blocks = [b for b in sym.split("\n\n") if b.strip()]
assert len(blocks) == len(insts), (len(blocks), len(insts))
src = open("path_tracer_amd/csrc/pt_path_kernel.hip").read().splitlines()
per = collections.defaultdict(collections.Counter)
for (addr, op), b in zip(insts, blocks):
    lines = b.strip().splitlines()
    # last "file:line:col" entry is the outermost frame = the kernel body
    loc = lines[-1]
    m = re.search(r":(\d+):\d+$", loc)
    ln = int(m.group(1)) if m else 0
    k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "mem"
    if op.startswith(("s_load", "global_", "buffer_", "ds_", "flat_", "scratch_")): k = "mem"
    per[ln][k] += 1
tot = collections.Counter()
for ln in sorted(per):
    k = per[ln]; tot.update(k)
    print(f"{ln:4d} v{k['valu']:4d} s{k['salu']:4d} m{k['mem']:3d} | {src[ln-1].strip()[:120] if 0 < ln <= len(src) else ''}")
print("total", dict(tot))
