#!/bin/bash
# r06b: K1sw r06 form inside the configs[4] pipeline (bench line), then SQ / TCC counters of the r05 (opts 3) and r06 (opts 9)
# tile kernels, separate passes, kernel-trace only
cd "$(dirname "$0")/../.."
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --config c5w --steps 6 --warmup 2 --no-general-extra 2>$O/c5w.err | tail -1 > $O/c5w_line.json
python -c "import json; d=json.load(open('$O/c5w_line.json')); print(json.dumps({'c5w_ms_per_step': d['ms_per_step'], 'value': d['value'], 'k1sw_avg_launch_ms': d['roofline']['avg_launch_ms'], 'frac': d['roofline']['frac'], 'mfma_frac': d['roofline'].get('frac_of_fp32_matrix_peak'), 'iters': d['config']['iterations_per_step'], 'ok': d['check']['ok']}))" | tee $O/c5w_summary.json
for form in 3 9; do
  kn=dense_symm_wide7_kernel; [ $form = 9 ] && kn=dense_symm_wide8_kernel
  i=0
  for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf $O/p
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p -- python scripts/k1sw_pmc_run.py $form > /dev/null 2>$O/p$i.err || tail -2 $O/p$i.err
    F=$(find $O/p -name '*counter_collection.csv' | head -1)
    [ -n "$F" ] && { echo "# form $form: $C"; python scripts/pmc_parse.py $F $kn; } | tee -a $O/sq_counters_form$form.txt
    [ "$C" = FETCH_SIZE ] && cp $F $O/fetch_$form.csv
    [ "$C" = WRITE_SIZE ] && cp $F $O/write_$form.csv
    [ $i = 1 ] && cp $F $O/mfma_$form.csv
    rm -rf $O/p
  done
done
python scripts/pmc_collect.py dense_symm_wide8_kernel 17213947904 xk_symmwide.hip,xk_common.h $O/k1sw_pmc_traffic.json FETCH=$O/fetch_9.csv WRITE=$O/write_9.csv MFMA=$O/mfma_9.csv B=8 "note=K1sw r06 form (opts = 9), 8 x 32768^2 fp32, P = 16, tile kernel standalone (scripts/k1sw_pmc_run.py 9); algorithmic bytes = upper triangles + panels in + out"
rm -f $O/*.csv
