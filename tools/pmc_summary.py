"""Summarises rocprofv3 --pmc passes (one counter per pass) into profiles/r01_pmc_hbm_traffic.csv and
profiles/r01_pmc_traffic.json.

    python tools/pmc_summary.py gpurun_out/pmc

expects gpurun_out/pmc/<CONFIG>_<COUNTER>/**/*counter_collection.csv (COUNTER = FETCH_SIZE | WRITE_SIZE).
Counter values are KB per dispatch.  Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE
reports 1/2 of the bytes of wide coalesced streaming reads -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
The kernels summarised stream their operands with 16-byte-per-lane loads (GEMM: global_load_lds dwordx4);
the barycentric walk's 64-byte record gathers are an uncalibrated access width (reported, flagged)."""
import collections, csv, glob, json, os, re, sys

root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))     # (cfg, kernel) -> counter -> [values]
for d in sorted(glob.glob(os.path.join(root, "*_*"))):
    m = re.match(r"(C\d)_(FETCH_SIZE|WRITE_SIZE)$", os.path.basename(d))
    if not m:
        continue
    cfg, ctr = m.groups()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            name = re.sub(r"^void ", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            vals[(cfg, name)][ctr].append(float(r["Counter_Value"]))
rows, js = [], collections.defaultdict(dict)
KEEP = ("gemm_minus_streamk_kernel", "rbf_eval", "bary_eval_kernel", "rbf_fill_kernel", "trsv_dataflow_kernel", "chol_trsm128_kernel",
        "chol_diag128_kernel", "cell_", "tree_")
for (cfg, name), c in sorted(vals.items()):
    if not name.startswith(KEEP):
        continue
    f, w = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
    n = max(len(f), len(w))
    af, aw = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    mf, mw = (max(f) if f else 0.0), (max(w) if w else 0.0)
    rows.append((cfg, name, n, af, aw, mf, mw))
    short = re.sub(r"<.*$", "", name)
    big = (2.0 * mf + mw) * 1024.0                 # the largest launch of this kernel (GEMM: the top-level update)
    prev = js[cfg].get(short)
    if prev is None or big > prev:
        js[cfg][short] = int(big)
os.makedirs("profiles", exist_ok=True)
with open("profiles/r01_pmc_hbm_traffic.csv", "w") as fo:
    fo.write("# rocprofv3 --pmc passes (one counter per pass, with --kernel-trace only), MI355X, round 1, final kernels.\n")
    fo.write("# values are KB per dispatch, RAW; corrected bytes = (2*FETCH + WRITE)*1024 (gfx950 FETCH_SIZE halves wide streaming reads)\n")
    fo.write("# columns: config,kernel,launches,avg_FETCH_KB,avg_WRITE_KB,max_FETCH_KB,max_WRITE_KB\n")
    for r in rows:
        fo.write("%s,%s,%d,%.1f,%.1f,%.1f,%.1f\n" % r)
js["_note"] = ("HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 of the LARGEST launch of the kernel, from "
               "profiles/r01_pmc_hbm_traffic.csv (separate --pmc passes; gfx950 FETCH_SIZE correction x2 applied; "
               "Infinity-Cache hits are included in these memory-side counters; bary gathers are an uncalibrated width)")
json.dump(js, open("profiles/r01_pmc_traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(js, indent=1, sort_keys=True))
