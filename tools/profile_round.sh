#!/bin/bash
# The measurements kept under profiles/ for one kernel version, in one go (on the GPU box, from the repo root):
#   bash tools/profile_round.sh <tag>          e.g. v10
# writes gpurun_out/<tag>/: bench.json (the bench line), stats/ (rocprofv3 --kernel-trace --stats of the same
# bench command), pmc/{f,w,p1..p4}/ (one rocprofv3 --pmc run per counter group, as MI355X_MICROARCH.md asks:
# no other trace domain alongside, within the per-block counter limits). tools/summarize_profiles.py turns
# stats/ and pmc/ into the small files committed under profiles/.
set -e
tag=${1:?tag}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 400 python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d "$out/stats" -o vm -- python3 "$R/bench.py" --no-cpu-baseline > "$out/stats.log" 2>&1
pass() { # name counters...
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$out/pmc/$name" -o vm -- python3 "$R/tools/prof_run.py" --spp 50 --batches 2 > "$out/pmc_$name.log" 2>&1
}
pass f FETCH_SIZE
pass w WRITE_SIZE
pass p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass p2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR
pass p3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass p4 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
cd "$R"
tail -c 600 "$out/bench.json"
ls "$out/stats" "$out/pmc"
