# DiT products alone (gpurun: bash tests/micro/prof_gemm.sh): builds tests/micro/gemm_bench against the library's gemm.hip and runs it -
# bit-identity of every tiling against the automatic choice, then us per launch per tile override (6404: four waves of 256 x 64)
set -e
R=${GRAFT_REPO_ROOT:-.}
cd $R
F="-O3 -std=c++17 --offload-arch=gfx950 -I fangyan_tts_amd/csrc -I include"
for f in fangyan_tts_amd/csrc/gemm.hip fangyan_tts_amd/csrc/runtime.hip tests/micro/gemm_bench.hip; do
  hipcc $F -c $f -o /tmp/$(basename $f .hip).o 2>/dev/null
done
hipcc --offload-arch=gfx950 /tmp/gemm_bench.o /tmp/gemm.o /tmp/runtime.o -o /tmp/gemm_bench
timeout -k 10 400 /tmp/gemm_bench $1
