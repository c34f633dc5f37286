RT=/opt/rocm-7.2.0/lib/llvm/lib/clang/22/lib/linux
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:allocator_may_return_null=1
echo "--- torch under asan preload"
LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so python -c "import torch; print(torch.cuda.is_available()); x=torch.zeros(4,device='cuda'); print(x.sum().item())" 2>&1 | tail -5; echo "rc ${PIPESTATUS[0]}"
echo "--- solver under asan"
MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_asan.so LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so python -c "
import __graft_entry__ as g
g.smoke()" 2>&1 | tail -5; echo "rc ${PIPESTATUS[0]}"
echo "--- pytest under asan"
MMP_LIB_PATH=$PWD/modelmesh_amd/lib/variants/libmmplace_asan.so LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so python -m pytest -p no:cacheprovider tests/test_abi_fuzz_gpu.py -m gpu -q -x 2>&1 | tail -8; echo "rc ${PIPESTATUS[0]}"
