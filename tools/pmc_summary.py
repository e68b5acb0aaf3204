"""Summarises rocprofv3 --pmc passes (one counter per pass, tools/profile_all.sh) into profiles/.

    python tools/pmc_summary.py gpurun_out/pmc [round-tag, default r02]

expects gpurun_out/pmc/<CONFIG>_<COUNTER>/**/*counter_collection.csv.  Writes
  profiles/<tag>_pmc_counters.json     every counter, per config and kernel: launches, sum, max over launches
  profiles/<tag>_pmc_hbm_traffic.csv   FETCH_SIZE / WRITE_SIZE (KB per dispatch, raw)
  profiles/<tag>_pmc_traffic.json      HBM-side bytes of the LARGEST launch of each kernel = (2*FETCH + WRITE)*1024
                                       (MI355X_MICROARCH.md, HBM section: gfx950 FETCH_SIZE reports 1/2 of the bytes of
                                       wide coalesced streaming reads; the kernels summarised stream with 16-byte-per-lane
                                       loads / global_load_lds dwordx4; the barycentric walk's 64-byte record gathers are
                                       an uncalibrated access width -- reported, flagged; Infinity-Cache hits are included
                                       in these memory-side counters)
  profiles/<tag>_pmc_mfma.json         MFMA-pipe busy fraction of the largest stream-K launch =
                                       SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
  profiles/<tag>_pmc_pipes.md          VALU utilisation of the sweep kernels from SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES etc.
"""
import collections, csv, glob, json, os, re, sys

root = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
vals = collections.defaultdict(lambda: collections.defaultdict(list))     # (cfg, kernel) -> counter -> [values per dispatch]
for d in sorted(glob.glob(os.path.join(root, "*_*"))):
    m = re.match(r"(C\d)_([A-Z0-9_]+)$", os.path.basename(d))
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

KEEP = ("gemm_minus_streamk_kernel", "rbf_eval", "bary_eval_kernel", "leafwalk_kernel", "lw_", "bary_walk_kernel", "bary_start_kernel", "bary_finish_kernel", "walk_pack_kernel", "rbf_fill_kernel", "trsv_dataflow_kernel", "chol_trsm128_kernel",
        "chol_diag128_kernel", "cell_", "tl_", "tree_", "unsort_", "jump_build_kernel", "centre_pack_kernel")
counters = collections.defaultdict(dict)
rows, traffic = [], collections.defaultdict(dict)
for (cfg, name), c in sorted(vals.items()):
    if not name.startswith(KEEP):
        continue
    counters[cfg][name] = {k: {"launches": len(v), "sum": sum(v), "max": max(v)} for k, v in c.items()}
    f, w = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
    if f or w:
        n = max(len(f), len(w))
        af, aw = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
        mf, mw = (max(f) if f else 0.0), (max(w) if w else 0.0)
        rows.append((cfg, name, n, af, aw, mf, mw))
        short = re.sub(r"<.*$", "", name)
        big = (2.0 * mf + mw) * 1024.0
        if big > traffic[cfg].get(short, 0):
            traffic[cfg][short] = int(big)

os.makedirs("profiles", exist_ok=True)
json.dump(counters, open(f"profiles/{tag}_pmc_counters.json", "w"), indent=1, sort_keys=True)
with open(f"profiles/{tag}_pmc_hbm_traffic.csv", "w") as fo:
    fo.write(f"# rocprofv3 --pmc passes (one counter per pass, with --kernel-trace only), MI355X, round {tag}, final kernels.\n")
    fo.write("# values are KB per dispatch, RAW; corrected bytes = (2*FETCH + WRITE)*1024 (gfx950 FETCH_SIZE halves wide streaming reads)\n")
    fo.write("# columns: config,kernel,launches,avg_FETCH_KB,avg_WRITE_KB,max_FETCH_KB,max_WRITE_KB\n")
    for r in rows:
        fo.write("%s,%s,%d,%.1f,%.1f,%.1f,%.1f\n" % r)
traffic["_note"] = (f"HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 of the LARGEST launch of the kernel, from "
                    f"profiles/{tag}_pmc_hbm_traffic.csv (separate --pmc passes; gfx950 FETCH_SIZE correction x2 applied; "
                    "Infinity-Cache hits are included in these memory-side counters; bary gathers are an uncalibrated width)")
json.dump(traffic, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1, sort_keys=True)

mfma = {}
lines = [f"# pipe utilisation from rocprofv3 --pmc passes ({tag}); one counter per pass, values summed over the launches of a bench run\n"]
for cfg, ks in counters.items():
    for name, c in ks.items():
        if name.startswith("gemm_minus_streamk_kernel<256") and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]["max"] / (1024.0 * c["GRBM_GUI_ACTIVE"]["max"] / 8.0)
            mfma.setdefault(cfg, {})["gemm_minus_streamk_kernel"] = {
                "mfma_busy": round(busy, 4),
                "source": f"profiles/{tag}_pmc_counters.json: max-over-launches SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
            lines.append(f"* {cfg} `{name}` (largest launch): MFMA pipe busy = {busy:.3f}\n")
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
            a, b = c["SQ_ACTIVE_INST_VALU"]["sum"], c["SQ_BUSY_CYCLES"]["sum"]
            g = c.get("GRBM_GUI_ACTIVE", {}).get("sum")
            extra = ""
            if g:
                # SQ_ACTIVE_INST_VALU counts quad-cycles (4 clk) a wave spends issuing VALU, summed over all SIMDs
                extra = f"; SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = {a * 4.0 / (1024.0 * g / 8.0):.3f}"
            w = c.get("SQ_WAVE_CYCLES", {}).get("sum")
            wa = c.get("SQ_WAIT_INST_ANY", {}).get("sum")
            if w and wa:
                extra += f"; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {wa / w:.3f}"
            lines.append(f"* {cfg} `{name}`: SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {a / b:.3f} ({c['SQ_ACTIVE_INST_VALU']['launches']} launches){extra}\n")
json.dump(mfma, open(f"profiles/{tag}_pmc_mfma.json", "w"), indent=1, sort_keys=True)
open(f"profiles/{tag}_pmc_pipes.md", "w").writelines(lines)
print(json.dumps(traffic, indent=1, sort_keys=True))
print("".join(lines))
