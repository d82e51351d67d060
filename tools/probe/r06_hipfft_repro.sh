# builds and runs the stand-alone hipFFT reproducer (tests/native/hipfft_repro.cpp: hipFFT + HIP runtime only, nothing of this repository)
# against the system ROCm's hipFFT / rocFFT and against the copies that ship inside the PyTorch wheel
R=${GRAFT_REPO_ROOT:-$(pwd)}
hipcc --offload-arch=gfx950 -O2 $R/tests/native/hipfft_repro.cpp -o /tmp/hipfft_repro -lhipfft 2>/dev/null
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
for embeds in explicit null; do
  echo "==== system ROCm ($(readlink -f /opt/rocm)), churn, $embeds embeds"; /tmp/hipfft_repro churn $embeds; echo "exit $?"
  echo "==== system ROCm, fresh process per shape, $embeds embeds"; /tmp/hipfft_repro fresh $embeds; echo "exit $?"
done
echo "==== PyTorch wheel's hipFFT / rocFFT ($TL), churn, explicit embeds"; LD_LIBRARY_PATH=$TL /tmp/hipfft_repro churn explicit; echo "exit $?"
echo "==== PyTorch wheel's hipFFT / rocFFT, fresh process per shape, explicit embeds"; LD_LIBRARY_PATH=$TL /tmp/hipfft_repro fresh explicit; echo "exit $?"
echo "==== PyTorch wheel's, churn, null embeds"; LD_LIBRARY_PATH=$TL /tmp/hipfft_repro churn null; echo "exit $?"
