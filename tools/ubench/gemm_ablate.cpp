// Ablation harness for the NT GEMM: build with -DTOAD_ABLATE_* and compare.
#include "../../toad_amd/csrc/capi.hip"
#include "../../toad_amd/csrc/gemm_f32.hip"
#include <vector>
int main(int argc, char **argv) {
    const int64_t M = 100000, K = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 512;
    float *X, *W, *Y, *b; void *ws; size_t wsb = toad_linear_ws_bytes(100000, 1024, 1024); hipMalloc(&ws, wsb);
    hipMalloc(&X, M * K * 4); hipMalloc(&W, N * K * 4); hipMalloc(&Y, M * N * 4); hipMalloc(&b, N * 4);
    std::vector<float> h(M * K);
    for (auto &v : h) v = ((rand() % 2001) - 1000) / 1000.0f;
    hipMemcpy(X, h.data(), M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), N * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, h.data(), N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) toad_linear_act_fwd_f32(X, W, b, Y, M, K, N, 1, 0.f, 0, ws, wsb, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; ++i) toad_linear_act_fwd_f32(X, W, b, Y, M, K, N, 1, 0.f, 0, ws, wsb, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%s M=%ld K=%ld N=%ld  %.1f us  %.1f TF/s\n", argv[0], (long)M, (long)K, (long)N, ms * 1e3, 2.0 * M * K * N / ms / 1e9);
    return 0;
}
