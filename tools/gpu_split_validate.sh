# r6: the half-precision radial scale on the library without packed fp32 instructions: suite, replays of the big forward, bench lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r06split}; mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/tests.log 2>&1; tail -5 $out/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
sha256sum hamgnn_amd/lib/libhamgnn_hip.so | tee $out/lib.sha256
for wl in sio2_10k mos2_1200; do
  timeout 300 python tools/gpu_deal_all.py save /tmp/h_$wl.pt --workload $wl 2>/dev/null | tail -1 | tee -a $out/replays.log
  timeout 900 python tools/gpu_deal_all.py check /tmp/h_$wl.pt --workload $wl --forwards ${2:-150} 2>/dev/null | tail -1 | tee -a $out/replays.log
done
HG_S_SPLIT=0 timeout 300 python tools/gpu_deal_all.py check /tmp/h_sio2_10k.pt --workload sio2_10k --forwards 3 2>/dev/null | tail -1 | tee -a $out/replays.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench.err; python -c "
import json; r=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print(round(r['value']), 'edges/s', round(r['ms_per_step'],2), 'ms frac', r['roofline']['frac'], 'complete', r.get('value_complete_programs'), 'acc', r.get('accuracy',{}).get('rel_max'))"
HG_S_SPLIT=0 timeout 600 python bench.py --no-cpu-baseline --no-accuracy --no-complete-pass 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HG_S_SPLIT=0', round(r['value']), 'edges/s', round(r['ms_per_step'],2), 'ms')"
