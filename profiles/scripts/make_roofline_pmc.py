"""profiles/rNN_* counter tables -> profiles/roofline_pmc.json (what bench.py's roofline block reads).

    python profiles/scripts/make_roofline_pmc.py r03b

Inputs (all committed under profiles/, all from ONE gpurun call, profiles/scripts/r03b.sh):
  rNN_calib_pmc_a.md   SQ counters of the VALU calibration microbenchmark (profiles/microbench/valu_rates.hip --calib):
                       every launch is 100 % VALU-issue-bound by construction (4 waves / SIMD, independent
                       instructions of ONE class), so  SIMD-cycles / SQ_INSTS_VALU  is the issue cost of that class
                       at saturation.  (rocprofv3 averages the tiny warm-up launch with the real one: x2 below.)
  rNN_pmc_a.md, rNN_pmc_b.md   the same counters + the per-type instruction counters on bench.py's kernels
  rNN_pmc_fetch_size.md, rNN_pmc_write_size.md   HBM-side traffic, separate passes (MI355X_MICROARCH.md: KiB units,
                       FETCH_SIZE doubled on gfx950 — 64 B counted per 128 B request)
  static mix of the kernels (profiles/scripts/valu_mix.py) for the instructions the per-type counters do not cover.

Issue-cycle model ("what bounds the raster kernels"):
  cycles = FMA_F32 c_fma + MUL_F32 c_mul + ADD_F32 (s_dpp c_dpp + (1 - s_dpp) c_add) + TRANS_F32 c_trans
           + INT32 c_int + (INSTS_VALU - those) c_other
  s_dpp   = static share of DPP adds among the kernel's v_add/v_sub,  c_other = static-mix-weighted cost of the
            uncounted classes (v_cmp, v_cndmask / v_min, v_mov, v_permlane*_swap),
  frac    = cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
SQ_ACTIVE_INST_VALU is NOT cycles: on the calibration launches it reads exactly 1 per ordinary VALU instruction and 2
per transcendental / permlane swap (a nominal quad-cycle weight), which is why rounds 1-2's "ACTIVE x 4 / SIMD-cycles"
could exceed 1.
"""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "profiles")
SIMDS, XCDS = 1024, 8


def table(path):
    rows = {}
    lines = [l for l in open(path).read().splitlines() if l.startswith("|")]
    hdr = [c.strip() for c in lines[0].strip("|").split("|")]
    for l in lines[2:]:
        c = [x.strip() for x in l.strip("|").split("|")]
        rows[c[0].strip("`")] = {h: float(v) for h, v in zip(hdr[1:], c[1:])}
    return rows


def find(rows, key):
    return next(v for k, v in rows.items() if key in k)


WORKLOADS = {      # key in roofline_pmc.json (what bench.py looks up) -> file suffix of the counter tables
    "metric": "", "street": "_street", "scene_graph_dropin": "_sg", "scene_graph_fused": "_sgf",
}
RASTER_KEYS = (("raster_bwd", "raster_bwd_short_kernel"), ("raster_fwd", "raster_fwd_pk_kernel"))


def kernel_table(a, b, fs, ws):
    """Every raster / sub-list kernel instance of a workload: time, counters, HBM bytes per launch (gfx950: FETCH_SIZE in
    KiB counting 64 B per 128 B request -> doubled; WRITE_SIZE in KiB)."""
    rows = []
    for name, ka in a.items():
        if not any(t in name for t in ("raster_", "list_window", "build_grec", "unpack_grads")):
            continue
        kb = b.get(name, {})
        f, w = fs.get(name, {}).get("FETCH_SIZE"), ws.get(name, {}).get("WRITE_SIZE")
        rows.append({"kernel": name[:100], "avg_us": ka.get("avg_us"), "calls": ka.get("dispatches"),
                     "SQ_INSTS_VALU": ka.get("SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": ka.get("SQ_ACTIVE_INST_VALU"),
                     "SQ_WAVE_CYCLES": ka.get("SQ_WAVE_CYCLES"), "SQ_INSTS_SALU": kb.get("SQ_INSTS_SALU"),
                     "fetch_KiB_raw": f, "write_KiB": w,
                     "hbm_bytes": None if f is None or w is None else int((2 * f + w) * 1024)})
    return sorted(rows, key=lambda r: -(r["avg_us"] or 0) * (r["calls"] or 1))


# timing slot of bench.py (`kernels_avg_ms`) -> substrings of the kernels whose launches the slot brackets
SLOT_KERNELS = {
    "project_fwd": ("project_fwd_kernel",), "project_bwd": ("project_bwd_kernel",), "sh_fwd": ("sh_fwd_kernel",),
    "sh_bwd": ("sh_bwd_",), "scan": ("scan_reduce_kernel", "scan_final_kernel", "scan_partials_kernel"),
    "map_isect": ("bin_count_kernel", "bin_emit_kernel"), "sort": ("rs_hist_kernel", "rs_scan_kernel", "rs_scatter_kernel"),
    "tile_bins": ("tile_bins32_kernel",), "pack_records": ("build_grec_kernel",), "unpack_grads": ("unpack_grads_kernel",),
    "raster_fwd": ("raster_fwd_pk_kernel",), "raster_bwd": ("raster_bwd_short_kernel", "raster_bwd_kernel"),
}


def slot_traffic(fs, ws):
    """HBM bytes per STEP of every timing slot, measured: sum over the slot's kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB
    x launches per step (the two counters come from two passes with their own step counts: launches of the forward
    raster kernel = steps of that pass)."""
    def steps(rows):
        return next(v["dispatches"] for k, v in rows.items() if "raster_fwd_pk_kernel" in k)
    sf, sw = steps(fs), steps(ws)
    out = {}
    for slot, keys in SLOT_KERNELS.items():
        f = sum(v["FETCH_SIZE"] * v["dispatches"] for k, v in fs.items() if any(t in k for t in keys)) / sf
        w = sum(v["WRITE_SIZE"] * v["dispatches"] for k, v in ws.items() if any(t in k for t in keys)) / sw
        out[slot] = {"hbm_bytes_per_step": int((2 * f + w) * 1024), "fetch_KiB_raw_per_step": round(f, 1),
                     "write_KiB_per_step": round(w, 1)}
    return out


def main(tag):
    cal = table(os.path.join(P, f"{tag}_calib_pmc_a.md"))
    cost = {}
    for k, v in cal.items():      # x2: two dispatches averaged, the first one (10 iterations) is ~0
        cyc = 2 * v["GRBM_GUI_ACTIVE"] / XCDS * SIMDS
        cost[k.split("(")[0]] = {"issue_cycles_per_inst": cyc / (2 * v["SQ_INSTS_VALU"]),
                                 "clock_ghz": 2 * v["GRBM_GUI_ACTIVE"] / XCDS / (2 * v["avg_us"] * 1e3),
                                 "active_inst_valu_per_inst": v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"]}
    c = {k: v["issue_cycles_per_inst"] for k, v in cost.items()}
    mix = json.loads(subprocess.check_output([sys.executable, os.path.join(P, "scripts", "valu_mix.py"), "--json"],
                                             stderr=subprocess.DEVNULL, text=True))
    sha_file = os.path.join(P, f"{tag}_raster_hip.sha256")
    measured_sha = open(sha_file).read().split()[0] if os.path.exists(sha_file) else None
    import hashlib
    here_sha = hashlib.sha256(open(os.path.join(ROOT, "street-gaussians-ns_amd", "csrc", "raster.hip"), "rb").read()).hexdigest()
    if measured_sha is not None and measured_sha != here_sha:
        print(f"WARNING: the counters were taken on another raster.hip ({measured_sha[:12]} vs {here_sha[:12]} here)")
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = {"_comment": __doc__.split("Inputs")[0].strip() + f"  Generated from profiles/{tag}_*; see the script for the "
           "formulas.", "round": tag, "git_head": head, "raster_hip_sha256": measured_sha or here_sha,
           "calibration": cost, "workloads": {}}
    for wkey, suf in WORKLOADS.items():
        files = [os.path.join(P, f"{tag}_pmc_{x}{suf}.md") for x in ("a", "b", "fetch_size", "write_size")]
        # the bench line of the same workload gives the evaluated-pair count (deterministic per scene): of this tag's
        # call, or — argv[2] — of the call whose bench set this counter campaign belongs to
        bench_file = os.path.join(P, f"{tag}_bench_{'default' if wkey == 'metric' else suf[1:]}.json.log")
        if not os.path.exists(bench_file) and BENCH_TAG:
            bench_file = os.path.join(P, f"{BENCH_TAG}_bench_{'default' if wkey == 'metric' else suf[1:]}.json.log")
        if not all(os.path.exists(f) for f in files):
            continue
        a, b, fs, ws = (table(f) for f in files)
        W = {"table": kernel_table(a, b, fs, ws), "source": [os.path.relpath(f, ROOT) for f in files]}
        bench = json.loads(open(bench_file).read().strip().splitlines()[-1]) if os.path.exists(bench_file) else None
        W["workload"] = bench["config"]["workload"] if bench else wkey
        if wkey == "metric":
            W["slots"] = slot_traffic(fs, ws)
            W["slots_source"] = f"profiles/{tag}_pmc_fetch_size.md, {tag}_pmc_write_size.md (every kernel of the step)"
        pairs = ((bench or {}).get("roofline", {}).get("walked") or {}).get("quadrant_pairs_evaluated_fwd")
        # the per-kernel roofline entries bench.py replays belong to the benchmark scene only: there ONE kernel instance
        # serves each raster slot.  On the other workloads a slot spans several instances (short- and long-walk halves of
        # the backward running concurrently; main pass, window passes and group walks of the scene graph), so they keep
        # the per-instance `table` above and nothing that could be mistaken for a per-slot figure.
        for name, key in (RASTER_KEYS if wkey == "metric" else ()):
            try:
                ka, kb, st = find(a, key), find(b, key), mix[name]["classes"]
                fetch_kib, write_kib = find(fs, key)["FETCH_SIZE"], find(ws, key)["WRITE_SIZE"]
            except StopIteration:
                continue
            tot = ka["SQ_INSTS_VALU"]
            counted = sum(kb[f"SQ_INSTS_VALU_{t}"] for t in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "CVT"))
            other = tot - counted
            s_dpp = st.get("dpp", 0) / max(1, st.get("dpp", 0) + st.get("add", 0))
            oth_cls = {"cmp": c["k_cmp"], "select": c["k_min"], "mov": c["k_mov"], "permlane": c["k_permswap"],
                       "pk": 5.0, "other": c["k_mov"]}     # v_pk_*: 5.0 cycles (profiles/r01d_profile.md, r02 notes)
            w = sum(st.get(k, 0) for k in oth_cls)
            c_other = sum(st.get(k, 0) * v for k, v in oth_cls.items()) / max(1, w)
            cycles = (kb["SQ_INSTS_VALU_FMA_F32"] * c["k_fma"] + kb["SQ_INSTS_VALU_MUL_F32"] * c["k_mul"] +
                      kb["SQ_INSTS_VALU_ADD_F32"] * (s_dpp * c["k_dpp"] + (1 - s_dpp) * c["k_mul"]) +
                      kb["SQ_INSTS_VALU_TRANS_F32"] * c["k_exp"] + (kb["SQ_INSTS_VALU_INT32"] + kb["SQ_INSTS_VALU_CVT"]) * c["k_mov"] +
                      other * c_other)
            simd_cycles = ka["GRBM_GUI_ACTIVE"] / XCDS * SIMDS
            W[name] = {
                "kernel": key, "avg_us_under_pmc": ka["avg_us"], "bound": "valu",
                "hbm_traffic_bytes": int((2 * fetch_kib + write_kib) * 1024),
                "hbm_source": f"profiles/{tag}_pmc_fetch_size{suf}.md ({fetch_kib:.0f} KiB raw, doubled) + "
                              f"profiles/{tag}_pmc_write_size{suf}.md ({write_kib:.0f} KiB)",
                "valu": {
                    "simds": SIMDS, "clock_ghz": ka["GRBM_GUI_ACTIVE"] / XCDS / (ka["avg_us"] * 1e3),
                    "valu_insts_per_launch": tot, "pairs_per_launch": pairs,
                    "valu_insts_per_pair": tot / pairs if pairs else None,
                    "issue_cycles_per_inst": cycles / tot,
                    "issue_cycle_frac_under_pmc": cycles / simd_cycles,
                    "counter_frac_calibrated": (ka["SQ_ACTIVE_INST_VALU"] / simd_cycles) / (cal_frac(cal, "k_mix")),
                    "dynamic_mix": {k: kb[f"SQ_INSTS_VALU_{k}"] for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "CVT")},
                    "uncounted": other, "c_other": c_other, "dpp_share_of_adds": s_dpp, "static_mix": st,
                    "source": f"profiles/{tag}_pmc_a{suf}.md, {tag}_pmc_b{suf}.md, {tag}_calib_pmc_a.md, scripts/valu_mix.py, "
                              "scripts/make_roofline_pmc.py",
                },
            }
            v = W[name]["valu"]
            print(wkey, name, "us %.1f  insts/pair %s  cycles/inst %.2f  issue frac %.3f  ACTIVE frac %.3f  hbm %.1f MB" % (
                ka["avg_us"], v["valu_insts_per_pair"] and round(v["valu_insts_per_pair"], 1), v["issue_cycles_per_inst"],
                v["issue_cycle_frac_under_pmc"], v["counter_frac_calibrated"], W[name]["hbm_traffic_bytes"] / 1e6))
        out["workloads"][wkey] = W
    json.dump(out, open(os.path.join(P, "roofline_pmc.json"), "w"), indent=1)
    print({k: round(v["issue_cycles_per_inst"], 2) for k, v in cost.items()})


def cal_frac(cal, k):
    v = find(cal, k)
    return v["SQ_ACTIVE_INST_VALU"] / (v["GRBM_GUI_ACTIVE"] / XCDS * SIMDS)


BENCH_TAG = sys.argv[2] if len(sys.argv) > 2 else None

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03b")
