"""Census of the loops of a kernel in a hipcc -S listing: per backward branch, the instruction mix between the label and the
branch (matrix instructions, LDS reads / writes, LDS-DMA requests, scalar spill traffic, waits).  Usage:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only openmpl_amd/csrc/h2_gemm.hip -o build_tmp/h2.s
  python tools/loop_census.py build_tmp/h2.s h2_stack_kernel [min_mfma]"""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kern + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
labels = {}
for i in range(start, end):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m: labels[m.group(1)] = i
def mix(a, b):
    c = dict(n=0, mfma=0, ds_read=0, ds_write=0, dma=0, readlane=0, writelane=0, vmcnt=0, lgkm=0, barrier=0, s_load=0, salu=0, valu=0, scratch=0)
    for l in lines[a:b]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"): continue
        op = t.split()[0]
        c["n"] += 1
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("ds_read") or op.startswith("ds_load"): c["ds_read"] += 1
        elif op.startswith("ds_write") or op.startswith("ds_store"): c["ds_write"] += 1
        elif "load_lds" in op: c["dma"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_readfirstlane"): c["readlane"] += 1
        elif op.startswith("v_writelane"): c["writelane"] += 1
        elif op == "s_waitcnt": c["vmcnt" if "vmcnt" in t else "lgkm"] += 1
        elif op == "s_barrier": c["barrier"] += 1
        elif op.startswith("s_load"): c["s_load"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("v_"): c["valu"] += 1
    return c
seen = []
for i in range(start, end):
    m = re.match(r"^\ts_cbranch_\w+ (\.LBB\d+_\d+)|^\ts_branch (\.LBB\d+_\d+)", lines[i])
    if not m: continue
    tgt = m.group(1) or m.group(2)
    if tgt in labels and labels[tgt] < i:
        c = mix(labels[tgt], i + 1)
        if c["mfma"] >= min_mfma: seen.append((labels[tgt], i, c))
for a, b, c in seen:
    print(f"lines {a}-{b}: " + " ".join(f"{k}={v}" for k, v in c.items() if v))
