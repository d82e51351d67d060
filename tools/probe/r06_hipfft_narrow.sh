# narrows the stand-alone hipFFT reproducer down: which EARLIER plan makes a new (16, 8, 32) plan wrong?
R=${GRAFT_REPO_ROOT:-$(pwd)}
hipcc --offload-arch=gfx950 -O2 $R/tests/native/hipfft_repro.cpp -o /tmp/hipfft_repro -lhipfft 2>/dev/null
for first in "16 16 16" "8 8 8" "16 8 24" "12 10 14" "8 64 16" "32 32 32" "24 16 8" "32 16 8"; do
  echo "--- $first, then 16 8 32"; /tmp/hipfft_repro seq explicit $first 16 8 32 | tail -1
done
echo "--- 32 16 8 then 24 16 8 then 16 8 32"; /tmp/hipfft_repro seq explicit 32 16 8 24 16 8 16 8 32 | tail -1
echo "--- all eight, then 16 8 32"; /tmp/hipfft_repro seq explicit 16 16 16 8 8 8 16 8 24 12 10 14 8 64 16 32 32 32 24 16 8 32 16 8 16 8 32 | tail -1
echo "--- first four, then 16 8 32"; /tmp/hipfft_repro seq explicit 16 16 16 8 8 8 16 8 24 12 10 14 16 8 32 | tail -1
echo "--- last four, then 16 8 32"; /tmp/hipfft_repro seq explicit 8 64 16 32 32 32 24 16 8 32 16 8 16 8 32 | tail -1
echo "--- 16 8 24 + 8 64 16, then 16 8 32"; /tmp/hipfft_repro seq explicit 16 8 24 8 64 16 16 8 32 | tail -1
echo "--- 8 64 16 + 32 32 32, then 16 8 32"; /tmp/hipfft_repro seq explicit 8 64 16 32 32 32 16 8 32 | tail -1
echo "--- 32 32 32 + 32 16 8, then 16 8 32"; /tmp/hipfft_repro seq explicit 32 32 32 32 16 8 16 8 32 | tail -1
echo "--- 16 8 32 twice"; /tmp/hipfft_repro seq explicit 16 8 32 16 8 32 | tail -1
echo "--- 16 16 16 after 16 8 32 16 16 8"; /tmp/hipfft_repro seq explicit 16 8 32 16 16 8 16 16 16 | tail -1
