// TEST INFRASTRUCTURE ONLY: runtime half of tests/emu/hip/hip_runtime.h (see the header comment there).
#include <hip/hip_runtime.h>

thread_local uint3_emu threadIdx;
thread_local uint3_emu blockIdx;
thread_local dim3 blockDim;
thread_local dim3 gridDim;

namespace hipemu {
BlockCtx* g_ctx = nullptr;
std::vector<unsigned char>* g_wave_big = nullptr;
thread_local unsigned t_linear = 0;

struct ThreadArg { unsigned tid; dim3 grid, block; LaunchArgsBase* body; };

static void* thread_main(void* p) {
    ThreadArg* a = static_cast<ThreadArg*>(p);
    t_linear = a->tid;
    blockDim = a->block;
    gridDim = a->grid;
    threadIdx.x = a->tid % a->block.x;
    threadIdx.y = (a->tid / a->block.x) % a->block.y;
    threadIdx.z = a->tid / (a->block.x * a->block.y);
    for (unsigned bz = 0; bz < a->grid.z; ++bz)
        for (unsigned by = 0; by < a->grid.y; ++by)
            for (unsigned bx = 0; bx < a->grid.x; ++bx) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                a->body->run();
                sync_block();   // nobody enters the next block while `static __shared__` is still in use
            }
    return nullptr;
}

void launch(dim3 grid, dim3 block, LaunchArgsBase* body) {
    unsigned n = block.x * block.y * block.z;
    BlockCtx ctx;
    ctx.nthreads = n;
    pthread_barrier_init(&ctx.block_bar, nullptr, n);
    unsigned nw = (n + 63) / 64;
    ctx.wave_bar.resize(nw);
    for (unsigned w = 0; w < nw; ++w) pthread_barrier_init(&ctx.wave_bar[w], nullptr, std::min(64u, n - w * 64));
    ctx.wave_buf.assign(nw * 64, 0);
    std::vector<unsigned char> big((size_t)nw * 64 * 64, 0);
    g_wave_big = &big;
    g_ctx = &ctx;
    std::vector<pthread_t> th(n);
    std::vector<ThreadArg> args(n);
    pthread_attr_t attr; pthread_attr_init(&attr); pthread_attr_setstacksize(&attr, 1 << 20);
    for (unsigned i = 0; i < n; ++i) { args[i] = {i, grid, block, body}; pthread_create(&th[i], &attr, thread_main, &args[i]); }
    for (unsigned i = 0; i < n; ++i) pthread_join(th[i], nullptr);
    pthread_attr_destroy(&attr);
    pthread_barrier_destroy(&ctx.block_bar);
    for (unsigned w = 0; w < nw; ++w) pthread_barrier_destroy(&ctx.wave_bar[w]);
    g_ctx = nullptr;
    g_wave_big = nullptr;
}
}  // namespace hipemu
