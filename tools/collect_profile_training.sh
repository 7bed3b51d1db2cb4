#!/bin/bash
# kernel table of the training step (Si-512, 6 steps incl. the first, compiling one): gpurun_out/$1/training_si512_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-train}; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/prof --output-format csv -- python tests/bench_training.py --workload si512 --steps 6 > $out/train_si512.log 2> $out/prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/training_si512_kernel_stats.csv
rm -rf $out/prof
grep "^step" $out/train_si512.log | tail -2
python - <<PY
import csv
rows = list(csv.DictReader(open("$out/training_si512_kernel_stats.csv")))
tot = sum(int(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print("total ms", tot / 1e6, "launches", calls)
def grp(pred):
    s = [r for r in rows if pred(r['Name'])]
    return round(sum(int(r['TotalDurationNs']) for r in s) / 1e6, 1), sum(int(r['Calls']) for r in s)
for name, pred in (("tp_wgrad", lambda n: n.startswith('tp_wgrad')), ("tp_is", lambda n: 'tp_is_kernel' in n), ("linear_wgrad", lambda n: n.startswith('linear_wgrad')),
                   ("block_gemm", lambda n: 'block_gemm' in n), ("library GEMM", lambda n: n.startswith('Cijk')), ("at::native", lambda n: 'at::native' in n),
                   ("copy/fill", lambda n: 'copyBuffer' in n or 'fillBuffer' in n)):
    print(name, grp(pred))
for r in rows[:28]: print(r['Calls'], round(int(r['TotalDurationNs']) / 1e6, 2), r['Name'][:110])
PY
