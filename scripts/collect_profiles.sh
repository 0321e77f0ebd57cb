#!/bin/bash
# Runs ON THE GPU BOX (from the repo root, under gpurun): bench.py + the rocprofv3 passes the profiles/ summaries are built from.
#   gpurun -- 'bash scripts/collect_profiles.sh gpurun_out/prof'
# then, back in the build container:  python scripts/build_profiles.py gpurun_out/prof r02
# Counter passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section); the program sits directly after `--`.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/${1:-gpurun_out/prof}
mkdir -p "$O"
python3 "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
cd /tmp && export TMPDIR=/tmp
A="--steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-profile"      # (13 forwards: one cold launch no longer carries 4-8 % of a kernel's average - VERDICT r4 weak #11)
F="--output-format csv"
timeout -k 10 250 rocprofv3 --kernel-trace --stats $F -d "$O/prof_kt" -o runc -- python3 "$R/bench.py" $A > "$O/prof_kt.log" 2>&1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE $F -d "$O/pmc_fetch" -o runc -- python3 "$R/bench.py" $A > "$O/pmc_fetch.log" 2>&1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE $F -d "$O/pmc_write" -o runc -- python3 "$R/bench.py" $A > "$O/pmc_write.log" 2>&1
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY $F -d "$O/pmc_sq" -o runc -- python3 "$R/bench.py" $A > "$O/pmc_sq.log" 2>&1
timeout -k 10 250 rocprofv3 --kernel-trace --stats $F -d "$O/prof_kt_f16" -o runc -- python3 "$R/bench.py" $A --precision f16 > "$O/prof_kt_f16.log" 2>&1
cd "$R"
TS2D_DBG=256 timeout -k 10 200 python3 scripts/gpu_ops_only.py split > "$O/phase_stamps.txt" 2>&1 || true
(cd scripts/probes && timeout -k 10 120 ./hbm_probe > "$O/hbm_probe.txt" 2>&1) || true
echo done
