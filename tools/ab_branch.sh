# decoder alone (tools/ab_branch.py), then the whole pipeline at 2 / 1 utterances in flight and BASELINE cfg 1, alternating
mkdir -p gpurun_out/r05b
timeout 600 python3 tools/ab_branch.py > gpurun_out/r05b/ab_branch.txt 2>&1
cat gpurun_out/r05b/ab_branch.txt | grep -v amdgpu.ids
run() { name=$1; shift; timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-rooflines "$@" > gpurun_out/r05b/$name.json 2> gpurun_out/r05b/$name.err; echo "$name $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r05b/$name.json | head -1)"; }
for i in 1 2 3; do
run i2_s0_$i --vocoder-side-streams 0
run i2_s2_$i --vocoder-side-streams 2
run i1_s0_$i --inflight 1 --vocoder-side-streams 0
run i1_s2_$i --inflight 1 --vocoder-side-streams 2
run c1_s0_$i --config 1 --vocoder-side-streams 0
run c1_s2_$i --config 1 --vocoder-side-streams 2
done
