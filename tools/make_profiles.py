"""Turn the rocprofv3 outputs of tools/gpu_profile_session.sh (gpurun_out/prof_<tag>/) into the tracked profiles/ files.
    python tools/make_profiles.py <round tag, e.g. r02>"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
G = "gpurun_out/prof_%s" % tag
def find(pat):
    c = glob.glob(os.path.join(G, pat), recursive=True)
    return c[0] if c else None
shutil.copy(find("trace/**/t_kernel_stats.csv"), "profiles/%s_bench_kernel_stats.csv" % tag)
open("profiles/%s_bench_n1.json" % tag, "w").write(open("%s/bench.json" % G).read().strip() + "\n")
out = open("profiles/%s_pmc_summary.txt" % tag, "w")
def P(*a): print(*a, file=out)
P("rocprofv3 PMC passes (each its own run, --kernel-trace only) over")
P("  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra")
P("i.e. the kernels the headline forward itself launches.  Values are per-launch averages over all launches of a kernel;")
P("x3_stack_kernel = ALL 52 GEMMs of the FPT block stack in one persistent launch (csrc/x3_gemm.hip).")
fetch, write = {}, {}
for p, desc in [("a", "SQ pass"), ("b", "LDS / L2 pass"), ("c", "FETCH_SIZE pass (kB; gfx950: x2 for wide streaming reads, MI355X_MICROARCH.md HBM section)"),
                ("d", "WRITE_SIZE pass (kB)")]:
    path = find("pmc_%s/**/p_counter_collection.csv" % p)
    if not path:
        continue
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(list)); d = collections.defaultdict(dict)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mpl::", "")
        if not any(k in name for k in ("x3_stack", "x3_gemm", "spt_kernel", "fuse_head", "split_rows", "row_stats")):
            continue
        key = (name, r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"])
        a[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[key][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    P("\n== pass %s: %s" % (p, desc))
    for key in a:
        ds = list(d[key].values())
        P("%s grid=%s wg=%s vgpr=%s launches=%d avg_dur_us=%.1f" % (key + (len(ds), sum(ds) / len(ds) / 1e3)))
        for c, v in sorted(a[key].items()):
            P("    %-28s %.6g" % (c, sum(v) / len(v)))
            if c == "FETCH_SIZE" and "x3_stack" in key[0]: fetch = (sum(v) / len(v), len(ds))
            if c == "WRITE_SIZE" and "x3_stack" in key[0]: write = (sum(v) / len(v), len(ds))
out.close()
if fetch and write:
    fb, wb = fetch[0] * 2 * 1024, write[0] * 1024
    json.dump({"kernel": "x3_stack_kernel", "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
               "traffic_bytes_per_launch": fb + wb,
               "source": "profiles/%s_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py itself, FETCH_SIZE x2 "
                         "gfx950 correction; one launch = the 52 GEMMs of the block stack)" % tag},
              open("profiles/%s_gemm_traffic.json" % tag, "w"), indent=1)
    print("traffic MB per launch", fb / 1e6, wb / 1e6)
