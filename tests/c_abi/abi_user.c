/* A plain C99 caller of the drop-in boundary (include/mi355cube.h): what a cgo / JNI / Rust `extern "C"` binding sees.
 * Compiled with -std=c99 -pedantic -Wall -Werror by tests/test_abi_cpu.py and linked against the product library; it takes the
 * address of every entry point it uses (the prototypes must parse as C, not only as C++), fills the descriptor structs by field
 * name, and calls what can be called without a device: the ABI version, the host-side planning entry points
 * (mi355_gemm_relayout_plan, mi355_gemm_tail_plan, mi355_pitched_row_bytes) and the NULL-context error paths. */
#include <stdio.h>
#include <string.h>

#include "mi355cube.h"

/* With a device: the hot path from C alone.  Operands of ones generated in HBM (mi355_fill_uniform with lo == hi), three products whose
 * every output must be exactly K -- [N][K] rhs, row-major rhs, transposed lhs (on all-ones operands the layouts only differ in how the
 * kernels walk them) -- read back and compared as bf16 bits; then the 1 Mi-element sum of config C1. */
#define CHECK(call) do { const int32_t rc_ = (call); if (rc_ != MI355_OK) { printf("%s -> %d (%s)\n", #call, (int)rc_, mi355_last_error(ctx)); return 1; } } while (0)
static int on_device(void)
{
    enum { M = 512, N = 768, K = 1024 };
    static uint16_t host[M * N];
    mi355_ctx *ctx = NULL;
    if (mi355_ctx_create(0, &ctx) != MI355_OK) { printf("ctx_create failed\n"); return 1; }
    void *a = NULL, *b = NULL, *c = NULL, *x = NULL, *sum = NULL, *ws = NULL;
    CHECK(mi355_alloc(ctx, (uint64_t)M * K * 2, &a));
    CHECK(mi355_alloc(ctx, (uint64_t)N * K * 2, &b));
    CHECK(mi355_alloc(ctx, (uint64_t)M * N * 2, &c));
    CHECK(mi355_fill_uniform(ctx, NULL, a, MI355_DTYPE_BF16, (uint64_t)M * K, 1, 1, 1.0f, 1.0f));
    CHECK(mi355_fill_uniform(ctx, NULL, b, MI355_DTYPE_BF16, (uint64_t)N * K, 1, 2, 1.0f, 1.0f));
    int bad = 0;
    for (int layout = 0; layout < 3; ++layout) {
        mi355_gemm_desc d;
        memset(&d, 0, sizeof d);
        d.m = M; d.n = N; d.k = K; d.batch = 1;
        d.trans_a = layout == 2; d.trans_b = layout == 0;
        d.lda = d.trans_a ? M : K; d.ldb = d.trans_b ? K : N; d.ldc = N;
        d.dtype_ab = MI355_DTYPE_BF16; d.dtype_c = MI355_DTYPE_BF16; d.algo = MI355_GEMM_ALGO_AUTO;
        CHECK(mi355_memset(ctx, NULL, c, 0xEE, (uint64_t)M * N * 2));
        CHECK(mi355_gemm(ctx, NULL, &d, a, b, c));
        CHECK(mi355_read(ctx, NULL, host, c, (uint64_t)M * N * 2));
        int wrong = 0;
        for (int i = 0; i < M * N; ++i) wrong += host[i] != 0x4480;          /* bf16(1024.0) */
        printf("gemm from C, layout %d: %d of %d outputs differ from K\n", layout, wrong, M * N);
        bad += wrong != 0;
    }
    const uint64_t n = 1u << 20;
    uint64_t ws_bytes = 0;
    CHECK(mi355_alloc(ctx, n * 4, &x));
    CHECK(mi355_alloc(ctx, 16, &sum));
    CHECK(mi355_reduce_workspace_bytes(ctx, n, &ws_bytes));
    CHECK(mi355_alloc(ctx, ws_bytes ? ws_bytes : 16, &ws));
    CHECK(mi355_fill_uniform(ctx, NULL, x, MI355_DTYPE_F32, n, 1, 3, 0.5f, 0.5f));
    CHECK(mi355_reduce_sum(ctx, NULL, x, MI355_DTYPE_F32, n, (float *)sum, ws, ws_bytes));
    float s = 0.f;
    CHECK(mi355_read(ctx, NULL, &s, sum, 4));
    printf("sum of 2^20 halves from C: %.1f\n", (double)s);
    bad += s != 524288.0f;
    CHECK(mi355_free(ctx, a)); CHECK(mi355_free(ctx, b)); CHECK(mi355_free(ctx, c));
    CHECK(mi355_free(ctx, x)); CHECK(mi355_free(ctx, sum)); CHECK(mi355_free(ctx, ws));
    CHECK(mi355_ctx_destroy(ctx));
    return bad;
}

int main(void)
{
    int failures = 0;
    const int32_t version = mi355_abi_version();
    printf("abi %d (header %d)\n", (int)version, (int)MI355_ABI_VERSION);
    if (version != MI355_ABI_VERSION) ++failures;

    /* the benchmark's descriptor (config C3: 8192^3 bf16, B stored [N][K]) and its row-major twin: nothing is re-laid out */
    mi355_gemm_desc d;
    memset(&d, 0, sizeof d);
    d.m = d.n = d.k = 8192; d.batch = 1;
    d.lda = d.ldb = d.ldc = 8192;
    d.dtype_ab = MI355_DTYPE_BF16; d.dtype_c = MI355_DTYPE_BF16;
    d.trans_b = 1; d.algo = MI355_GEMM_ALGO_AUTO;
    int32_t ra = -1, rb = -1;
    if (mi355_gemm_relayout_plan(&d, &ra, &rb) != MI355_OK || ra != 0 || rb != 0) ++failures;
    d.trans_b = 0;
    if (mi355_gemm_relayout_plan(&d, &ra, &rb) != MI355_OK || ra != 0 || rb != 0) ++failures;
    /* lhs stored [K][M] (lhs^T . grad_out): staged natively as well */
    d.trans_a = 1;
    if (mi355_gemm_relayout_plan(&d, &ra, &rb) != MI355_OK || ra != 0 || rb != 0) ++failures;
    printf("relayout plan of the transposed-lhs 8192^3: a %d b %d\n", (int)ra, (int)rb);

    /* no context: every entry point refuses, none crashes */
    if (mi355_gemm(NULL, NULL, &d, NULL, NULL, NULL) == MI355_OK) ++failures;
    if (mi355_sync(NULL, NULL) == MI355_OK) ++failures;
    /* ... except the host-side planning ones, which need none: the transposed-lhs 8192^3 runs on the 256x256 kernel */
    int32_t algo = -1;
    if (mi355_gemm_select(NULL, &d, &algo) != MI355_OK || algo != MI355_GEMM_ALGO_LP_256W4) ++failures;
    if (mi355_gemm_select(NULL, NULL, &algo) == MI355_OK) ++failures;

    /* function pointers: the prototypes are usable as C types */
    int32_t (*gemm)(mi355_ctx *, mi355_stream, const mi355_gemm_desc *, const void *, const void *, void *) = mi355_gemm;
    int32_t (*count)(int32_t *) = mi355_device_count;
    int32_t n = -1;
    const int32_t rc = count(&n);
    printf("device_count rc %d n %d; gemm entry %s\n", (int)rc, (int)n, gemm ? "bound" : "missing");
    if (rc == MI355_OK && n >= 1) failures += on_device();
    printf(failures ? "C ABI user: %d FAILURES\n" : "C ABI user ok\n", failures);
    return failures;
}
