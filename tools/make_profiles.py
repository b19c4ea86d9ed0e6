"""Turn the rocprofv3 outputs of tools/gpu_profile_session.sh (gpurun_out/prof_<tag>/) into the tracked profiles/ files.
    python tools/make_profiles.py <round tag, e.g. r03> [<suffix of an extra workload, e.g. cmu_v8_bf16>]

For the headline workload it writes  profiles/<tag>_bench_kernel_stats.csv, _bench_n1.json, _pmc_summary.txt and
_gemm_traffic.json.  The per-launch figures of the dominant kernel are taken from the HEADLINE launches only: the rows
are grouped by (kernel, grid, workgroup, VGPRs) and the group with the most launches wins (ties: the largest grid) --
the same process also launches the kernel once on the 64-pose parity batch, which must not be averaged in or, worse,
be the one reported (round 2 did exactly that)."""
import collections, csv, glob, json, os, shutil, sys

if len(sys.argv) > 2 and sys.argv[1] == "--check":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from openmpl_amd import build as _b
    cur = _b.source_hash()
    for f in sorted(glob.glob("profiles/%s_*gemm_traffic.json" % sys.argv[2])):
        h = json.load(open(f)).get("srchash")
        print("%-50s %s" % (f, "HEAD sources" if h == cur else "OTHER sources (%s...)" % str(h)[:12]))
    sys.exit(0)
tag = sys.argv[1]
extra = sys.argv[2] if len(sys.argv) > 2 else None
G = "gpurun_out/prof_%s" % tag + ("_" + extra if extra else "")
stem = "profiles/%s_%s" % (tag, extra + "_" if extra else "")


def find(pat):
    c = glob.glob(os.path.join(G, pat), recursive=True)
    return c[0] if c else None


def short(name):
    return name.split("(")[0].replace("void ", "").replace("mpl::", "")


ks = find("trace/**/t_kernel_stats.csv")
if ks:
    shutil.copy(ks, stem + ("kernel_stats.csv" if extra else "bench_kernel_stats.csv"))
bj = None
if os.path.exists("%s/bench.json" % G):
    txt = open("%s/bench.json" % G).read().strip()
    open(stem + ("bench.json" if extra else "bench_n1.json"), "w").write(txt + "\n")
    try:
        bj = json.loads(txt)
    except Exception:
        bj = None
cmdline = open("%s/cmd.txt" % G).read().strip() if os.path.exists("%s/cmd.txt" % G) else \
    "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"

out = open(stem + "pmc_summary.txt", "w")


def P(*a):
    print(*a, file=out)


P("rocprofv3 PMC passes (each its own run, --kernel-trace only) over")
P("  " + cmdline)
P("i.e. the kernels the forward itself launches.  Values are per-launch averages over the launches of one")
P("(kernel, grid, workgroup, VGPR) group; the group with the most launches is the timed workload (HEADLINE), smaller")
P("groups of the same kernel are the parity check / warm-up shapes of the same process.")
PASSES = [("a", "SQ pass"), ("b", "LDS / L2 pass"),
          ("c", "FETCH_SIZE pass (kB; gfx950: x2 for wide streaming reads, MI355X_MICROARCH.md HBM section)"),
          ("d", "WRITE_SIZE pass (kB)")]
head = {}          # kernel short name -> {counter: per-launch mean of the headline group}
head_meta = {}
for p, desc in PASSES:
    path = find("pmc_%s/**/p_counter_collection.csv" % p)
    if not path:
        continue
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    d = collections.defaultdict(dict)
    for r in rows:
        if "mpl::" not in r["Kernel_Name"]:
            continue                                   # torch's own copy / fill kernels are not part of the forward
        name = short(r["Kernel_Name"])
        key = (name, int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]))
        a[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[key][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    best = {}
    for key in a:
        n = len(d[key])
        if key[0] not in best or (n, key[1]) > (len(d[best[key[0]]]), best[key[0]][1]):
            best[key[0]] = key
    P("\n== pass %s: %s" % (p, desc))
    for key in sorted(a, key=lambda k: (k[0], -len(d[k]))):
        ds = list(d[key].values())
        is_head = best[key[0]] == key
        P("%s grid=%d wg=%d vgpr=%d launches=%d avg_dur_us=%.1f%s" % (key + (len(ds), sum(ds) / len(ds) / 1e3,
                                                                           "   <-- HEADLINE" if is_head else "")))
        for c, v in sorted(a[key].items()):
            P("    %-28s %.6g" % (c, sum(v) / len(v)))
            if is_head:
                head.setdefault(key[0], {})[c] = sum(v) / len(v)
        if is_head:
            head_meta.setdefault(key[0], {}).update(grid=key[1], wg=key[2], vgpr=key[3], launches=len(ds))
            head[key[0]].setdefault("dur_us", {})[p] = sum(ds) / len(ds) / 1e3

# ---- derived figures of every kernel of the forward
P("\n== derived (headline groups)")
derived = {}
for name, c in sorted(head.items()):
    m = dict(head_meta[name])
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        m["fetch_bytes"] = c["FETCH_SIZE"] * 2 * 1024          # gfx950 correction: wide streaming reads are tallied at 1/2
        m["write_bytes"] = c["WRITE_SIZE"] * 1024
        m["traffic_bytes"] = m["fetch_bytes"] + m["write_bytes"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        m["tcc_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "GRBM_GUI_ACTIVE" in c:
        m["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8 / (c["dur_us"]["a"] * 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            m["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["clock_ghz"] * c["dur_us"]["a"] * 1e3)
    if "SQ_INSTS_VALU" in c and c.get("SQ_INSTS_MFMA"):
        m["valu_per_mfma"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in c:
                m[k.lower() + "_frac"] = c[k] / c["SQ_WAVE_CYCLES"]
    derived[name] = m
    P(name, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in m.items()}))
out.close()

# ---- traffic of the dominant kernel (bench.py reads the newest rNN_gemm_traffic.json for roofline.traffic)
dom = None
if bj and not extra:
    dom = bj.get("roofline", {}).get("kernel")
cands = [n for n in derived if "traffic_bytes" in derived[n] and (dom is None or n.startswith(dom))]
if cands:
    name = max(cands, key=lambda n: derived[n]["traffic_bytes"] * derived[n]["launches"])
    m = derived[name]
    alg = None
    if bj:
        alg = bj.get("roofline", {}).get("algorithmic_bytes_per_launch")
    j = {"kernel": name.split("<")[0], "kernel_instance": name, "grid": m["grid"], "launches_averaged": m["launches"],
         "fetch_bytes_per_launch": m["fetch_bytes"], "write_bytes_per_launch": m["write_bytes"],
         "traffic_bytes_per_launch": m["traffic_bytes"], "algorithmic_bytes_per_launch": alg,
         "traffic_ratio": (m["traffic_bytes"] / alg) if alg else None, "tcc_hit_rate": m.get("tcc_hit_rate"),
         # the library the passes ran on: bench.py reports roofline.traffic only while the sources still hash to this
         "srchash": (open("%s/srchash.txt" % G).read().strip() if os.path.exists("%s/srchash.txt" % G) else None),
         "source": "%spmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the bench command itself, HEADLINE "
                   "launch group only, FETCH_SIZE x2 gfx950 correction)" % stem}
    json.dump(j, open(stem + "gemm_traffic.json", "w"), indent=1)
    # a profile set taken on other kernel sources than the tree's is evidence for THOSE sources -- say so here, and
    # `python tools/make_profiles.py --check <tag>` lists every committed set of a round whose hash is not the tree's
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from openmpl_amd import build as _b
        if j["srchash"] and j["srchash"] != _b.source_hash():
            print("WARNING: this session ran on library sources %s..., the tree hashes to %s...: stale evidence for HEAD"
                  % (j["srchash"][:12], _b.source_hash()[:12]))
    except Exception as e:
        print("(source hash not compared: %r)" % (e,))
    print("traffic MB per launch: fetch %.1f write %.1f ratio %s hit %s" % (m["fetch_bytes"] / 1e6, m["write_bytes"] / 1e6,
                                                                           j["traffic_ratio"], j["tcc_hit_rate"]))
