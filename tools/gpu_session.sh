#!/bin/bash
# One parameterised GPU session script (replaces the per-session gpu_r3*.sh files of round 3).  Run on the GPU box through gpurun:
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <tag> <stage> [<stage> ...]'
# Every stage writes gpurun_out/<tag>_<stage>.log (merged back into the build container); stages are independent and each is bounded by `timeout`.
#   checks:<pytest -k expr>   per-op parity checks (tests/test_hip_ops_gpu.py) selected by the expression
#   pytest                    the whole GPU suite (-m gpu)
#   smoke                     __graft_entry__.smoke()
#   bench[:<args>]            python bench.py <args>
#   probe:<tool>[:<args>]     python tools/<tool>.py <args>
#   oproof                    tools/op_roofline.py fp32 (per-(op, shape) table of one image pass)
#   pmc:<dtype>:<kernel>      three separate --pmc passes over `bench.py --roofline-only --dtype <dtype>` + tools/pmc_summary.py
#   stats:<dtype>             rocprofv3 --kernel-trace --stats of the same roofline command
#   headline                  all-16-tile parity of configs[2] (PF_HEADLINE_ALL=1)
#   ab:<V1>+<V2>+...          tools/image_ab.py (interleaved whole-image A/B) over the variants; a variant is NAME=VAL[,NAME=VAL], the empty variant = default tree
#   lib:<path>                PF_LIB_PATH for the stages that follow (A/B library builds: make -C patchfusion_amd/csrc variant NAME=.. DEFS=..; lib: resets)
#   env:VAR=VALUE             export for the stages that follow (env:VAR= unsets)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; TAG=$1; shift
for st in "$@"; do
  name=${st%%:*}; arg=""; [[ "$st" == *:* ]] && arg=${st#*:}
  log=$O/${TAG}_$(echo "$st" | tr -c 'A-Za-z0-9_\n' '_' | cut -c1-60).log
  case $name in
    make) ( make -C patchfusion_amd/csrc $arg 2>&1 | tail -3 ) > $log 2>&1 ;;
    env) if [[ -z "${arg#*=}" ]]; then unset "${arg%%=*}"; else export "$arg"; fi; echo "== $st"; continue ;;
    checks) ( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "$arg" -s 2>&1 | tail -40 ) > $log 2>&1 ;;
    pytest) ( timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -20 ) > $log 2>&1 ;;
    smoke) ( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > $log 2>&1 ;;
    bench) ( timeout 900 python bench.py $(echo $arg | tr ',' ' ') ) > $log 2> $log.err ;;
    probe) tool=${arg%%:*}; targs=""; [[ "$arg" == *:* ]] && targs=${arg#*:}
           ( timeout 900 python tools/$tool.py $(echo $targs | tr ',' ' ') ) > $log 2>&1 ;;
    oproof) ( timeout 600 python tools/op_roofline.py fp32 $O/${TAG}_op_roofline_fp32.md $O/${TAG}_op_roofline_fp32.json ) > $log 2>&1 ;;
    pmc) dt=${arg%%:*}; kern=${arg#*:}; P="python bench.py --roofline-only --dtype $dt"
         ( timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_pmc_${dt}_a -o p -- $P ) > $O/${TAG}_pmc_${dt}_a.log 2>&1
         ( timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_${dt}_b -o p -- $P ) > $O/${TAG}_pmc_${dt}_b.log 2>&1
         ( timeout 300 rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/${TAG}_pmc_${dt}_c -o p -- $P ) > $O/${TAG}_pmc_${dt}_c.log 2>&1
         python tools/pmc_summary.py $dt $kern $O/${TAG}_pmc_dominant_${dt}.json $O/${TAG}_pmc_${dt}_a $O/${TAG}_pmc_${dt}_b $O/${TAG}_pmc_${dt}_c > $log 2>&1
         rm -f $O/${TAG}_pmc_*/p_kernel_trace.csv $O/${TAG}_pmc_*/*/*kernel_trace.csv ;;
    stats) ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_$arg -o ro -- python bench.py --roofline-only --dtype $arg ) > $log 2> $log.err
           find $O/${TAG}_stats_$arg -name '*kernel_trace.csv' -delete ;;
    ab) IFS='+' read -r -a vs <<< "$arg+"; ( timeout 1800 python tools/image_ab.py --steps 4 --rounds 3 "${vs[@]}" ) > $O/${TAG}_image_ab.md 2> $log ;;
    lib) if [[ -z "$arg" ]]; then unset PF_LIB_PATH; else export PF_LIB_PATH="$PWD/$arg"; fi; echo "== $st"; continue ;;
    headline) ( PF_HEADLINE_ALL=1 timeout 900 python -m pytest tests/test_headline_parity_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > $log 2>&1 ;;
    *) echo "unknown stage $st" ;;
  esac
  echo "== $st"; tail -c 3000 $log
done
