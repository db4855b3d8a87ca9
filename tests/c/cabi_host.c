/* A plain C host of the C ABI (no torch, no C++): what a C/C++ maintainer's binding of include/qatiger.h looks like.
 * Build (tests/test_cabi_c_host.py does it):
 *   gcc -std=c99 tests/c/cabi_host.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Lqa-tiger_amd -lqatiger_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -o cabi_host
 * Runs y = relu(x W^T + b) through qt_gemm_f32 and qt_gemm_x3 on device buffers and checks it against a host loop;
 * also checks that a bad argument returns an error code and a message instead of launching. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "qatiger.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(void) {
    const int M = 300, N = 512, K = 256;
    float *x = malloc(sizeof(float) * M * K), *w = malloc(sizeof(float) * N * K), *b = malloc(sizeof(float) * N);
    float *y = malloc(sizeof(float) * M * N), *ref = malloc(sizeof(float) * M * N);
    unsigned s = 12345u;
    for (int i = 0; i < M * K; ++i) { s = s * 1664525u + 1013904223u; x[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    for (int i = 0; i < N * K; ++i) { s = s * 1664525u + 1013904223u; w[i] = (((s >> 8) & 0xffff) / 32768.0f - 1.0f) * 0.1f; }
    for (int i = 0; i < N; ++i) b[i] = 0.01f * (float)(i % 7 - 3);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double acc = b[n];
            for (int k = 0; k < K; ++k) acc += (double)x[m * K + k] * (double)w[n * K + k];
            ref[m * N + n] = acc > 0.0 ? (float)acc : 0.0f;
        }
    float *dx, *dw, *db, *dy;
    CHECK(hipMalloc((void**)&dx, sizeof(float) * M * K)); CHECK(hipMalloc((void**)&dw, sizeof(float) * N * K));
    CHECK(hipMalloc((void**)&db, sizeof(float) * N)); CHECK(hipMalloc((void**)&dy, sizeof(float) * M * N));
    CHECK(hipMemcpy(dx, x, sizeof(float) * M * K, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw, w, sizeof(float) * N * K, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b, sizeof(float) * N, hipMemcpyHostToDevice));
    printf("qt_version %d\n", qt_version());
    int failures = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int rc = pass == 0
            ? qt_gemm_f32(NULL, 0, 1, M, N, K, 1.0f, dx, K, dw, K, dy, N, db, NULL, 0, 1, NULL, 0, 1.0f, 0.0f, 0ull, 0, 0, 1, 0, 0, 0)
            : qt_gemm_x3(NULL, 0, 0, 1, M, N, K, 1.0f, dx, K, dw, K, dy, N, db, NULL, 0, 1, NULL, 0, 1.0f, 0.0f, 0ull, 0, 0, NULL, 1, 0, 0, 0, NULL, NULL, NULL, NULL, 0);
        if (rc != 0) { fprintf(stderr, "gemm pass %d: rc=%d %s\n", pass, rc, qt_last_error()); return 3; }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(y, dy, sizeof(float) * M * N, hipMemcpyDeviceToHost));
        double worst = 0.0, scale = 0.0;
        for (int i = 0; i < M * N; ++i) { double d = fabs((double)y[i] - ref[i]); if (d > worst) worst = d; if (fabs(ref[i]) > scale) scale = fabs(ref[i]); }
        printf("%s: max rel err %.2e\n", pass == 0 ? "qt_gemm_f32" : "qt_gemm_x3 (3x fp16)", worst / scale);
        if (worst / scale > 1e-5) ++failures;
    }
    int rc = qt_gemm_f32(NULL, 0, 1, -1, N, K, 1.0f, dx, K, dw, K, dy, N, NULL, NULL, 0, 0, NULL, 0, 1.0f, 0.0f, 0ull, 0, 0, 1, 0, 0, 0);
    printf("bad argument -> rc %d (%s)\n", rc, qt_last_error());
    if (rc == 0) ++failures;
    hipFree(dx); hipFree(dw); hipFree(db); hipFree(dy);
    free(x); free(w); free(b); free(y); free(ref);
    printf(failures ? "FAIL\n" : "OK\n");
    return failures ? 1 : 0;
}
