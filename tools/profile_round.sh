#!/bin/bash
# The measurements kept under profiles/ for one kernel version and one configuration, in one go (on the GPU box, from
# the repo root):
#   bash tools/profile_round.sh <tag> [config=c2] [arithmetic=exact] [spp]       e.g. v15 c3 fast
# writes gpurun_out/<tag>_<config>_<arithmetic>/: bench.json (the bench line), stats/ (rocprofv3 --kernel-trace --stats
# of the same bench command), count.log (instrumented launch: counters + algorithmic bytes), pmc/{f,w,p1..p5}/ (one
# rocprofv3 --pmc run per counter group, as MI355X_MICROARCH.md asks: no other trace domain alongside, within the
# per-block counter limits). tools/summarize_profiles.py turns stats/ and pmc/ into kernel_stats.csv and pmc.json there (the databases are
# deleted: too large to return), which are copied under profiles/.
set -e
tag=${1:?tag}
config=${2:-c2}
arith=${3:-exact}
spp=${4:-}
R=$(pwd)
out=$R/gpurun_out/${tag}_${config}_${arith}
mkdir -p "$out"
sppflag=""
[ -n "$spp" ] && sppflag="--spp $spp"
timeout -k 10 400 python3 bench.py --config $config --arithmetic $arith > "$out/bench.json" 2> "$out/bench.err"
timeout -k 10 300 python3 tools/prof_run.py --config $config --arithmetic $arith $sppflag --batches 2 --count > "$out/count.log" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d "$out/stats" -o vm -- python3 "$R/bench.py" --config $config --arithmetic $arith --no-cpu-baseline > "$out/stats.log" 2>&1
pass() { # name counters...
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" -d "$out/pmc/$name" -o vm -- python3 "$R/tools/prof_run.py" --config $config --arithmetic $arith $sppflag --batches 2 > "$out/pmc_$name.log" 2>&1
}
pass f FETCH_SIZE
pass w WRITE_SIZE
pass p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass p2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR
pass p3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass p4 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
pass p5 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32
cd "$R"
# the rocpd databases are tens of MiB each and gpurun_out/ returns at most 64 MiB: summarise here, keep the summaries
ab=$(grep algorithmic_bytes "$out/count.log" | awk '{print $2}')
python3 tools/summarize_profiles.py stats "$out/stats/vm_results.db" "$out/kernel_stats.csv" > "$out/summary.log" 2>&1
python3 tools/summarize_profiles.py pmc "$out/pmc" "$out/pmc.json" "$ab" "rocprofv3 --pmc <pass> -- python3 tools/prof_run.py --config $config --arithmetic $arith $sppflag --batches 2" >> "$out/summary.log" 2>&1
rm -rf "$out/stats" "$out/pmc"
tail -c 300 "$out/bench.json"
