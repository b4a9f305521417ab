// TEST INFRASTRUCTURE ONLY: runtime half of tests/emu/hip/hip_runtime.h (see the header comment there).
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#if !defined(__x86_64__)
#error "the HIP emulation's fiber switch is written for x86-64"
#endif
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#endif
#endif

thread_local uint3_emu threadIdx;
thread_local uint3_emu blockIdx;
thread_local dim3 blockDim;
thread_local dim3 gridDim;

// void hipemu_switch(void** save_sp, void* load_sp): park the callee-saved state on the current stack, continue on another
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(".text\n"
    ".globl hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");

extern "C" char __start_hipemu_lds[] __attribute__((weak));
extern "C" char __stop_hipemu_lds[] __attribute__((weak));

namespace hipemu {
std::vector<unsigned char>* g_wave_big = nullptr;
thread_local unsigned t_linear = 0;
thread_local uint64_t* t_wave_buf = nullptr;

namespace {
#ifndef HIPEMU_STACK_BYTES
#define HIPEMU_STACK_BYTES (512 << 10)
#endif
constexpr size_t kStack = HIPEMU_STACK_BYTES;      // per lane; the sanitizer builds of the most unrolled kernels need more (tools/emu_tsan_check.sh)
enum Reason { kNone = 0, kWave = 1, kBlock = 2, kDone = 3 };

struct Fiber {
    void* sp;
    char* stack;
    int reason;
    unsigned tid;
    uint3_emu tidx;
};
struct Wave {
    void* sched_sp;
    Fiber* cur;
    LaunchArgsBase* body;
};
thread_local Wave t_wave;
pthread_barrier_t g_block_bar;
const bool g_poison_lds = getenv("HIPEMU_POISON_LDS") != nullptr;

[[noreturn]] void fiber_entry() {
    t_wave.body->run();
    t_wave.cur->reason = kDone;
    hipemu_switch(&t_wave.cur->sp, t_wave.sched_sp);
    abort();                                  // a finished lane is never resumed
}
void fiber_reset(Fiber& f) {
#ifdef HIPEMU_ASAN
    __asan_unpoison_memory_region(f.stack, kStack);   // tools/emu_asan_check.sh: a reused stack carries no stale redzones
#endif
    // the slot holding the entry address is 16-byte aligned, so the entry function starts with the ABI's stack alignment
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack - 64) & ~uintptr_t(15);
    void** slot = reinterpret_cast<void**>(top);
    slot[0] = reinterpret_cast<void*>(&fiber_entry);
    for (int i = 1; i <= 6; ++i) slot[-i] = nullptr;   // rbp, rbx, r12 .. r15
    f.sp = slot - 6;
    f.reason = kNone;
}
void yield(int reason) {
    Fiber* f = t_wave.cur;
    f->reason = reason;
    hipemu_switch(&f->sp, t_wave.sched_sp);
}

struct WaveArg { unsigned wave, lanes, block_threads; dim3 grid, block; LaunchArgsBase* body; uint64_t* slots; };

void* wave_main(void* p) {
    WaveArg* a = static_cast<WaveArg*>(p);
    blockDim = a->block;
    gridDim = a->grid;
    t_wave_buf = a->slots;
    t_wave.body = a->body;
    std::vector<Fiber> lanes(a->lanes);
    char* stacks = static_cast<char*>(mmap(nullptr, kStack * a->lanes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (stacks == MAP_FAILED) { perror("hipemu: mmap"); abort(); }
    for (unsigned l = 0; l < a->lanes; ++l) {
        Fiber& f = lanes[l];
        f.stack = stacks + (size_t)l * kStack;
        f.tid = a->wave * 64 + l;
        f.tidx.x = f.tid % a->block.x;
        f.tidx.y = (f.tid / a->block.x) % a->block.y;
        f.tidx.z = f.tid / (a->block.x * a->block.y);
    }
    for (unsigned bz = 0; bz < a->grid.z; ++bz)
        for (unsigned by = 0; by < a->grid.y; ++by)
            for (unsigned bx = 0; bx < a->grid.x; ++bx) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                if (g_poison_lds) {                 // every wave waits until wave 0 has filled LDS with NaN patterns
                    if (a->wave == 0 && __start_hipemu_lds) memset(__start_hipemu_lds, 0xff, __stop_hipemu_lds - __start_hipemu_lds);
                    pthread_barrier_wait(&g_block_bar);
                }
                for (Fiber& f : lanes) fiber_reset(f);
                for (;;) {
                    unsigned live = 0, at_block = 0;
                    for (Fiber& f : lanes) {
                        if (f.reason == kDone) continue;
                        t_wave.cur = &f;
                        t_linear = f.tid;
                        threadIdx = f.tidx;
                        hipemu_switch(&t_wave.sched_sp, f.sp);
                        if (f.reason != kDone) { ++live; at_block += f.reason == kBlock; }
                    }
                    if (live == 0) break;
                    if (at_block) {
                        if (at_block != live) { fprintf(stderr, "hipemu: divergent __syncthreads in wave %u\n", a->wave); abort(); }
                        pthread_barrier_wait(&g_block_bar);
                    }
                }
                pthread_barrier_wait(&g_block_bar);   // nobody enters the next block while `static __shared__` is still in use
            }
    munmap(stacks, kStack * a->lanes);
    return nullptr;
}
}  // namespace

void sync_wave() { yield(kWave); }
void sync_block() { yield(kBlock); }

void launch(dim3 grid, dim3 block, LaunchArgsBase* body) {
    const unsigned n = block.x * block.y * block.z;
    const unsigned nw = (n + 63) / 64;
    pthread_barrier_init(&g_block_bar, nullptr, nw);
    std::vector<uint64_t> slots((size_t)nw * 64, 0);
    std::vector<unsigned char> big((size_t)nw * 64 * 64, 0);
    g_wave_big = &big;
    std::vector<pthread_t> th(nw);
    std::vector<WaveArg> args(nw);
    pthread_attr_t attr; pthread_attr_init(&attr); pthread_attr_setstacksize(&attr, 1 << 20);
    for (unsigned w = 0; w < nw; ++w) {
        args[w] = {w, std::min(64u, n - w * 64), n, grid, block, body, slots.data() + (size_t)w * 64};
        pthread_create(&th[w], &attr, wave_main, &args[w]);
    }
    for (unsigned w = 0; w < nw; ++w) pthread_join(th[w], nullptr);
    pthread_attr_destroy(&attr);
    pthread_barrier_destroy(&g_block_bar);
    g_wave_big = nullptr;
}
}  // namespace hipemu
