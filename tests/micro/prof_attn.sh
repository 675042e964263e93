# DiT attention alone (gpurun: bash tests/micro/prof_attn.sh): builds tests/micro/attn_bench against the library sources and runs
# it for the round-5 kernel (attn_dit.hip) and the round-2 kernels (FY_ATTN_V1=1) at the benchmark's shape and a ragged / long one.
set -e
R=${GRAFT_REPO_ROOT:-.}
cd $R
F="-O3 -std=c++17 --offload-arch=gfx950 -I fangyan_tts_amd/csrc -I include"
hipcc $F -fno-slp-vectorize -c fangyan_tts_amd/csrc/attn_dit.hip -o /tmp/attn_dit.o
for f in fangyan_tts_amd/csrc/attn.hip fangyan_tts_amd/csrc/gemv32.hip fangyan_tts_amd/csrc/runtime.hip tests/micro/attn_bench.hip; do
  hipcc $F -c $f -o /tmp/$(basename $f .hip).o 2>/dev/null
done
hipcc --offload-arch=gfx950 /tmp/attn_bench.o /tmp/attn.o /tmp/gemv32.o /tmp/runtime.o /tmp/attn_dit.o -o /tmp/attn_bench
[ -n "$1" ] || for shape in "400 16" "650 8" "150 2" "256 16"; do
  for v in 0 1; do
    echo "== T nseq = $shape, FY_ATTN_V1=$v"
    FY_ATTN_V1=$v timeout -k 10 120 /tmp/attn_bench $shape
  done
done
# ablations of the round-5 kernel at the benchmark's shape (results are wrong by construction; only the time is read)
if [ "$1" = abl ]; then
for a in 1 2 3 4 5; do
  hipcc $F -fno-slp-vectorize -DAT2_ABL=$a -c fangyan_tts_amd/csrc/attn_dit.hip -o /tmp/attn_dit_$a.o
  hipcc --offload-arch=gfx950 /tmp/attn_bench.o /tmp/attn.o /tmp/gemv32.o /tmp/runtime.o /tmp/attn_dit_$a.o -o /tmp/attn_bench_$a
  echo "== ablation $a (1 no exp2, 2 no MFMAs, 3 no K/V loads in the loop, 4 = 3 + no barrier, 5 no maxima)"
  timeout -k 10 120 /tmp/attn_bench_$a 400 16 | tail -2
done
fi
# per-workgroup phase stamps of the round-5 kernel
if [ "$1" = stamps ]; then
  hipcc $F -fno-slp-vectorize -DAT2_STAMPS -c fangyan_tts_amd/csrc/attn_dit.hip -o /tmp/attn_dit_s.o
  hipcc $F -DAT2_STAMPS -c tests/micro/attn_bench.hip -o /tmp/attn_bench_s.o 2>/dev/null
  hipcc --offload-arch=gfx950 /tmp/attn_bench_s.o /tmp/attn.o /tmp/gemv32.o /tmp/runtime.o /tmp/attn_dit_s.o -o /tmp/attn_bench_s
  for shape in "400 16" "650 8"; do timeout -k 10 120 /tmp/attn_bench_s $shape | grep -v "^check"; done
fi
