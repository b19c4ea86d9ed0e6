"""Turn the rocprofv3 outputs of tools/gpu_profile_session.sh (under gpurun_out/) into the tracked profiles/ files.
    python tools/make_profiles.py <round tag, e.g. r01> <session id, e.g. 11>"""
import collections, csv, json, os, shutil, sys
tag, sid = sys.argv[1], sys.argv[2]
G = "gpurun_out"
shutil.copy("%s/prof%s/r01_kernel_stats.csv" % (G, sid), "profiles/%s_bench_kernel_stats.csv" % tag)
open("profiles/%s_bench_n1.json" % tag, "w").write(open("%s/bench%s.log" % (G, sid)).read().strip().splitlines()[-1] + "\n")
out = open("profiles/%s_gemm_pmc_summary.txt" % tag, "w")
def P(*a): print(*a, file=out)
P("rocprofv3 PMC passes (each its own run, --kernel-trace only) over `python tools/gemm_ab.py 544` = the four FPT GEMM shapes")
P("at M=4096, D=544 (65 launches each), and over `python tools/spt_ab.py` for the fused SPT kernel.")
P("kernel = mpl::x3_gemm_kernel<EPI, LN, NPASS, NST, DBG> (MPL_GEMM_X3=1: the split-operand GEMMs of the default fp32 path);")
P("values are per-launch averages; GRBM_GUI_ACTIVE is summed over the 8 XCDs.")
fetch, write = {}, {}
for p, desc, filt in [("a", "SQ pass", "x3_gemm"), ("b", "LDS / L2 pass", "x3_gemm"),
                      ("c", "FETCH_SIZE pass (kB; doubled below for the traffic figure, MI355X_MICROARCH.md HBM section)", "x3_gemm"),
                      ("d", "WRITE_SIZE pass (kB)", "x3_gemm"), ("e", "fused SPT kernel", "spt_kernel")]:
    path = "%s/pmc%s%s/p_counter_collection.csv" % (G, sid, p)
    if not os.path.exists(path):
        continue
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(list)); d = {}
    for r in rows:
        if filt not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("(")[0].replace("void mpl::", ""), r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"])
        a[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[(key, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    P("\n== pass %s: %s" % (p, desc))
    for key in a:
        ds = [v for (k, _), v in d.items() if k == key]
        P("%s grid=%s wg=%s vgpr=%s launches=%d avg_dur_us=%.1f" % (key + (len(ds), sum(ds) / len(ds) / 1e3)))
        for c, v in sorted(a[key].items()):
            P("    %-28s %.6g" % (c, sum(v) / len(v)))
            if c == "FETCH_SIZE": fetch[key[0] + key[1]] = (sum(v) / len(v), len(ds))
            if c == "WRITE_SIZE": write[key[0] + key[1]] = (sum(v) / len(v), len(ds))
out.close()
if fetch and write:
    fb = sum(v * n for v, n in fetch.values()) / sum(n for _, n in fetch.values()) * 2 * 1024
    wb = sum(v * n for v, n in write.values()) / sum(n for _, n in write.values()) * 1024
    # algorithmic bytes of the four launches of a block at M=4096, D=544: A (fp32) + split W (3 bf16 parts on 144 of
    # 136 columns) + C (+ residual read for proj / fc2)
    M_, D_ = 4096, 544
    alg = [M_ * k * 4 + n * k * 6 * 144 / 136 + M_ * n * 4 * (2 if res else 1)
           for k, n, res in ((D_, 3 * D_, 0), (D_, D_, 1), (D_, 2 * D_, 0), (2 * D_, D_, 1))]
    json.dump({"kernel": "x3_gemm_kernel", "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
               "traffic_bytes_per_launch": fb + wb, "algorithmic_bytes_per_launch": sum(alg) / 4,
               "source": "profiles/%s_gemm_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/gemm_ab.py, FETCH_SIZE x2 gfx950 correction; stand-alone GEMMs, i.e. the QKV launch still writes its packed output here)" % tag},
              open("profiles/%s_gemm_traffic.json" % tag, "w"), indent=1)
    print("traffic MB", fb / 1e6, wb / 1e6)
