#!/bin/bash
# usage: tools/pmc_gemm.sh OUTDIR M N K tA tB mode   -- three separate --pmc passes (kernel-trace only)
set -e
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE"
P3="FETCH_SIZE TCC_HIT"
P4="WRITE_SIZE TCC_MISS TCC_REQ"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -o pass$i -- python3 tools/one_gemm.py "$@" > $OUT/pass$i.log 2>&1 || { tail -5 $OUT/pass$i.log; exit 1; }
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/pass*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name'] and 'split' not in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print('%-28s last-dispatch %.4g  (n=%d)' % (k, v[-1], len(v)))
dur = []
for f in sorted(glob.glob(out + '/pass1_kernel_trace.csv')):
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('kernel us (pass1):', dur)
PY
