#!/bin/bash
# live parity soak against the CPU oracle on tracks / weight seeds no fixture covers (+ one probe: conv without the per-stage weight DMA)
set -o pipefail
mkdir -p gpurun_out
if [ "$1" = "i" ]; then
  echo "== conv, probe build without the per-stage weight DMA (timing only, outputs are wrong by construction)" > gpurun_out/r03_soak_i.log
  AC_LIB=libaudiocut_hip_nowdma.so timeout -k 10 200 python tools/conv_pf_bench.py 32 >> gpurun_out/r03_soak_i.log 2>&1
  echo "== conv, product build" >> gpurun_out/r03_soak_i.log
  timeout -k 10 200 python tools/conv_pf_bench.py 32 >> gpurun_out/r03_soak_i.log 2>&1
  echo "== soak I: v2.2_mdd, energy-gate VAD" >> gpurun_out/r03_soak_i.log
  timeout -k 10 1000 python tools/parity_soak.py "60,401,71,c1_sine_silence" "90,402,72,c1_sine_silence" "120,403,73,c2_song" "75,404,74,vocal_like" "100,405,75,voice_with_rests" "150,406,76,c2_song" "45,407,77,c1_sine_silence" >> gpurun_out/r03_soak_i.log 2>&1
elif [ "$1" = "k" ]; then
  echo "== soak K: longer tracks; Silero mode on burst / silence tracks (where the burst-calibrated synthetic network does fire)" > gpurun_out/r03_soak_k.log
  timeout -k 10 1150 python tools/parity_soak.py "240,421,91,c2_song" "120,423,93,c1_sine_silence,10" "150,424,94,c1_sine_silence,11" "200,425,95,voice_with_rests" "180,426,96,vocal_like" >> gpurun_out/r03_soak_k.log 2>&1
elif [ "$1" = "m" ]; then
  echo "== soak M: every generator again, half of the tracks with the Silero network" > gpurun_out/r03_soak_m.log
  timeout -k 10 1150 python tools/parity_soak.py "97,441,111,c2_song" "133,442,112,c2_song,15" "88,443,113,vocal_like,16" "111,444,114,vocal_like" "123,445,115,voice_with_rests,17" "77,446,116,voice_with_rests" "66,447,117,c1_sine_silence,18" "101,448,118,c1_sine_silence" >> gpurun_out/r03_soak_m.log 2>&1
elif [ "$1" = "n" ]; then
  echo "== soak N: final code, eight more tracks" > gpurun_out/r03_soak_n.log
  timeout -k 10 1150 python tools/parity_soak.py "84,451,121,c1_sine_silence,19" "130,452,122,c1_sine_silence" "105,453,123,c2_song" "71,454,124,c2_song,20" "95,455,125,vocal_like" "118,456,126,voice_with_rests" "52,457,127,c1_sine_silence,21" "160,458,128,c2_song" >> gpurun_out/r03_soak_n.log 2>&1
elif [ "$1" = "o" ]; then
  echo "== soak O: two long tracks (6 min of the long-form generator, 5 min of the song generator)" > gpurun_out/r03_soak_o.log
  timeout -k 10 1150 python tools/parity_soak.py "360,461,131,c5_long_form" "300,462,132,c2_song" >> gpurun_out/r03_soak_o.log 2>&1
else
  echo "== soak J: Silero network as the chunked VAD (soxr-HQ-specification resampler)" > gpurun_out/r03_soak_j.log
  timeout -k 10 1100 python tools/parity_soak.py "60,411,81,c1_sine_silence,3" "90,412,82,c2_song,4" "75,413,83,vocal_like,5" "120,414,84,voice_with_rests,6" "60,415,85,c2_song,7" "80,416,86,c1_sine_silence,8" >> gpurun_out/r03_soak_j.log 2>&1
fi
echo rc=$?
