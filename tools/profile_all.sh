cd /root/repo
bash tools/profile_round.sh r06 2>&1 | tail -40
for c in 8 16 32 64 128; do FIR_CH=$c python tools/fir_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ch $c: /"; done | tee gpurun_out/r06/fir_channels.log
python tools/bench_extra.py > gpurun_out/r06/bench_extra.jsonl 2> gpurun_out/r06/bench_extra.err; tail -3 gpurun_out/r06/bench_extra.err
