// Fiber scheduler of the host emulation (see hip/hip_runtime.h).  TEST INFRASTRUCTURE.
#include <hip/hip_runtime.h>

namespace emu {
thread_local dim3 t_idx, b_idx, b_dim, g_dim;
alignas(64) char dyn_lds[160 * 1024];
char xchg[1024][16];

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber { ucontext_t ctx; char *stack = nullptr; bool done = false, waiting = false, wwaiting = false; dim3 tid; };
std::vector<Fiber> fibers;
ucontext_t sched_ctx;
int cur = -1;
const std::function<void()> *cur_body = nullptr;

void trampoline()
{
    (*cur_body)();
    fibers[cur].done = true;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}
}  // namespace

void barrier()
{
    fibers[cur].waiting = true;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}

// rendezvous of the 64 lanes of the calling lane's WAVE (what the cross-lane instructions -- shuffles, MFMA -- are on the device):
// the waves of a workgroup need not execute the same number of them (exact.hip.h: matrix waves and epilogue waves)
void wave_barrier()
{
    fibers[cur].wwaiting = true;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
    const unsigned nt = block.x * block.y * block.z;
    if (fibers.size() < nt) fibers.resize(nt);          // (never shrunk: a Fiber dropped from the vector would take its stack's pointer with it)
    for (auto &f : fibers) if (!f.stack) f.stack = (char *)std::malloc(kStack);
    cur_body = &body;
    g_dim = grid; b_dim = block;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        for (unsigned t = 0; t < nt; t++) {
            Fiber &f = fibers[t];
            f.done = f.waiting = f.wwaiting = false;
            f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, trampoline, 0);
        }
        for (;;) {
            unsigned alive = 0;
            for (unsigned t = 0; t < nt; t++) {
                Fiber &f = fibers[t];
                if (f.done || f.waiting || f.wwaiting) continue;
                cur = (int)t; t_idx = f.tid; b_idx = dim3(bx, by, bz);
                swapcontext(&sched_ctx, &f.ctx);
            }
            for (unsigned t = 0; t < nt; t++) if (!fibers[t].done) alive++;
            if (!alive) break;
            // a wave whose live lanes all sit at a wave rendezvous goes on; only when none does, every live fiber sits at the
            // workgroup's barrier (or the kernel is wrong): release them (exited threads do not take part)
            bool wave_released = false;
            for (unsigned w0 = 0; w0 < nt; w0 += 64) {
                bool all = true, any = false;
                for (unsigned t = w0; t < nt && t < w0 + 64; t++) { if (fibers[t].done) continue; any = true; all = all && fibers[t].wwaiting; }
                if (any && all) { for (unsigned t = w0; t < nt && t < w0 + 64; t++) fibers[t].wwaiting = false; wave_released = true; }
            }
            if (wave_released) continue;
            for (unsigned t = 0; t < nt; t++) {
                if (!fibers[t].done && fibers[t].wwaiting) { std::fprintf(stderr, "emu: lane %u waits for its wave while the workgroup is at a barrier\n", t); std::abort(); }
                fibers[t].waiting = false;
            }
        }
    }
}
}  // namespace emu
