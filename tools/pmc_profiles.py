"""Developer tool: turn the rocprofv3 --pmc passes collected by tools/pmc_collect.sh into the two summaries under profiles/.

usage: python tools/pmc_profiles.py gpurun_out/r02_pmc r02
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, so it is doubled (calibrated here on gemv_u / gemv_t, whose algorithmic read is exactly
the 819 MB panel); WRITE_SIZE is taken as is.
"""
import collections, csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Counter_Name"] == "SQ_WAVE_CYCLES":
            durs[name].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in vals.items()}
want = ["kff_sym_kernel", "kff_sym_combine_kernel", "gemv_u_kernel", "gemv_t_kernel", "precond_z_kernel", "grad_kff_kernel", "grad_kff_gram_kernel",
        "select_step_kernel"]
hbm = {}
for k in want:
    if k in avg and "FETCH_SIZE" in avg[k]:
        f, w = avg[k]["FETCH_SIZE"], avg[k].get("WRITE_SIZE", 0.0)
        hbm[k] = {"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "fetch_bytes_corrected_x2": 2 * 1024 * f, "write_bytes": 1024 * w,
                  "hbm_bytes_per_launch": 2 * 1024 * f + 1024 * w}
json.dump({
    "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace --output-format csv -- python3 tools/pmc_run.py  [tools/pmc_collect.sh]",
    "workload": f"N=100000 D=8 M=1024 rbf fp64, MI355X, {tag} (default precision level)",
    "correction": "MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 64 B per 128-B request on gfx950 for wide coalesced reads -> doubled; WRITE_SIZE exact. Units KB.",
    "note": "The x2 calibration holds for the 16-B/lane streaming reads of gemv_u/gemv_t (2*405 MB = 810 MB vs 819 MB algorithmic). For kff_sym_kernel the reads are scalar-cache refills and row operands, the writes are the partial slabs (Prow/Pcol); treat its figure as an upper estimate. Infinity-Cache hits are included in FETCH_SIZE.",
    "kernels": hbm}, open(f"profiles/{tag}_pmc_hbm_traffic.json", "w"), indent=1)
sq = {}
for k in want:
    if k in avg:
        d = {c: v for c, v in avg[k].items() if c.startswith("SQ_") or c.startswith("GRBM")}
        if d:
            sq[k] = d
            if "SQ_INSTS_VALU_FMA_F64" in d:  # executed fp64 arithmetic of one launch: wave-instructions x 64 lanes, an fma = 2 flop
                d["executed_flop_per_launch"] = 64.0 * (2.0 * d["SQ_INSTS_VALU_FMA_F64"] + d.get("SQ_INSTS_VALU_ADD_F64", 0.0) + d.get("SQ_INSTS_VALU_MUL_F64", 0.0))
                d["valu_insts_per_launch"] = d.get("SQ_INSTS_VALU", 0.0)
json.dump({
    "command": "rocprofv3 --pmc <SQ counters> --kernel-trace --output-format csv -- python3 tools/pmc_run.py (two passes: instruction mix; cycles)  [tools/pmc_collect.sh]",
    "workload": f"N=100000 D=8 M=1024 rbf fp64, MI355X, {tag} (default precision level)",
    "kernels": sq}, open(f"profiles/{tag}_pmc_sq_counters.json", "w"), indent=1)
for k, d in hbm.items():
    print(f"{k:28s} fetch_raw {d['FETCH_SIZE_KB_raw']/1e3:9.1f} MB  write {d['WRITE_SIZE_KB_raw']/1e3:8.1f} MB  hbm/launch {d['hbm_bytes_per_launch']/1e6:9.1f} MB")
for k, d in sq.items():
    if "SQ_ACTIVE_INST_VALU" in d and "SQ_BUSY_CYCLES" in d:
        print(f"{k:28s} VALU insts {d.get('SQ_INSTS_VALU', 0):.3e}  active_valu/busy {d['SQ_ACTIVE_INST_VALU']/d['SQ_BUSY_CYCLES']:.2f}")
