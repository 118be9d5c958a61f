"""Copy the summaries of one tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked) and refresh
   profiles/hbm_traffic.json (read by bench.py for roofline.traffic).   usage: python tools/collect_profiles.py r02b "note" """
import json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
for a, b in (("driver_cmd_bench.json", "driver_cmd_bench.json"), ("bench_level4.json", "level4_bench.json"), ("bench_flat.json", "flat_bench.json"), ("bench_wfc_dr_8192.json", "wfc_dr_8192_bench.json"),
             ("bench_level4_quad.json", "level4_quad_layout_bench.json"), ("kernel_stats.csv", "level4_kernel_stats.csv"),
             ("bench_wfc_dr_8192_quad.json", "wfc_dr_8192_quad_layout_bench.json"), ("bench_level4_8192.json", "level4_8192_bench.json"),
             ("bench_level4_32768.json", "level4_32768_bench.json"), ("bench_flat_16384.json", "flat_16384_bench.json")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, f"{tag}_{b}"))
hdr = (f"# rocprofv3 PMC summary, {tag}: {note}; 4096 envs on level4, per-launch means; collected by tools/profile_round.sh\n"
       "# separate passes: --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU | --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE\n"
       f"#   -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline ; kernel trace ({tag}_level4_kernel_stats.csv): rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline (the default 300 + 30 steps)\n"
       "# FETCH_SIZE / WRITE_SIZE in KB as reported; summarised by tools/pmc_summary.py\n")
body = open(os.path.join(src, "pmc_summary.txt")).read()
open(os.path.join(dst, f"{tag}_level4_pmc_summary.txt"), "w").write(hdr + body)
val = {}
for ln in body.splitlines():
    m = re.match(r"void pgtt::physics_kernel<0.*?\s(FETCH_SIZE|WRITE_SIZE)\s+launches=\s*\d+ mean=([\d.e+]+)", ln)
    if m:
        val[m.group(1)] = float(m.group(2)) * 1024
tp = os.path.join(dst, "hbm_traffic.json")
t = json.load(open(tp))
# gfx950 correction (MI355X_MICROARCH.md "HBM"; calibrated on this library's own access pattern with tools/probes/traffic_calib.hip -
# 4-byte-per-lane coalesced SoA rows, 256 MiB read and written: FETCH_SIZE reports exactly 1/2 of the bytes read, WRITE_SIZE is exact)
FETCH_CORR, WRITE_CORR = 2.0, 1.0
t["level4_4096"] = {"physics_bytes_per_launch": FETCH_CORR * val["FETCH_SIZE"] + WRITE_CORR * val["WRITE_SIZE"],
                    "fetch_bytes": FETCH_CORR * val["FETCH_SIZE"], "write_bytes": WRITE_CORR * val["WRITE_SIZE"],
                    "raw_counters_KB": {"FETCH_SIZE": val["FETCH_SIZE"] / 1024, "WRITE_SIZE": val["WRITE_SIZE"] / 1024},
                    "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/{tag}_level4_pmc_summary.txt), KB * 1024, FETCH x 2 (gfx950 correction, "
                            "calibrated with tools/probes/traffic_calib.hip: 131 082 KB reported for 262 144 KB read, WRITE exact). These are L2 <-> fabric requests and include "
                            "Infinity-Cache hits: the 8 XCD L2s are not coherent with each other and start every launch cold, so each launch re-reads its state rows (1x) AND "
                            "the 0.8 MB terrain table once per XCD (6.4 MB, the one-forward reset launch shows the same fixed 8 MB) from the Infinity Cache; writes are "
                            "16-byte pieces of 128-byte lines per wave (4 envs per wave in the hex layout), counted as 64-byte requests"}
json.dump(t, open(tp, "w"), indent=1)
print("collected", tag, t["level4_4096"]["physics_bytes_per_launch"])
