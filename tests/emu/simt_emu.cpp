/* simt_emu.cpp -- functional SIMT emulator behind csrc/simt.h's X3_EMU mode (TESTS ONLY; see simt.h).
 * One ucontext fiber per GPU thread; 64 consecutive threads are a wave; the scheduler runs fibers round-robin and
 * switches at rendezvous points (ballot / shuffle / workgroup barrier).  Threads that have returned count as
 * inactive lanes, exactly like exited lanes on the hardware. */
#define X3_EMU 1
#include "../../x3_compressor_amd/csrc/simt.h"

#include <ucontext.h>
#include <vector>
#include <stdexcept>

x3emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {
struct Wave {
	unsigned size = 0, arrived = 0, finished = 0;
	unsigned long gen = 0;
	int pred[X3_WAVE];
	uint32_t val[X3_WAVE], res[X3_WAVE];
	bool here[X3_WAVE];
	uint64_t mask = 0;
};
struct Fiber {
	ucontext_t ctx;
	std::vector<char> stack;
	bool done = false;
	unsigned wave = 0, lane = 0;
	x3emu_dim3 tid;
};
std::vector<Fiber> fibers;
std::vector<Wave> waves;
ucontext_t sched_ctx;
int cur = -1;
unsigned blk_arrived = 0, blk_finished = 0;
unsigned long blk_gen = 0;
void (*kern_fn)(void *) = nullptr;
void *kern_arg = nullptr;

void yield_to_sched() { swapcontext(&fibers[cur].ctx, &sched_ctx); }

void wave_complete(Wave &w)
{
	uint64_t m = 0;
	for (unsigned l = 0; l < w.size; l++) {
		if (w.here[l] && w.pred[l]) m |= (uint64_t)1 << l;
		w.res[l] = w.val[l];
		w.here[l] = false;
	}
	w.mask = m;
	w.arrived = 0;
	w.gen++;
}

void wave_rendezvous(int pred, uint32_t val)
{
	Fiber &f = fibers[cur];
	Wave &w = waves[f.wave];
	w.pred[f.lane] = pred;
	w.val[f.lane] = val;
	w.here[f.lane] = true;
	w.arrived++;
	const unsigned long g = w.gen;
	if (w.arrived + w.finished == w.size) wave_complete(w);
	else while (w.gen == g) yield_to_sched();
}

void fiber_main()
{
	kern_fn(kern_arg);
	Fiber &f = fibers[cur];
	f.done = true;
	Wave &w = waves[f.wave];
	w.finished++;
	if (w.arrived && w.arrived + w.finished == w.size) wave_complete(w);
	blk_finished++;
	if (blk_arrived && blk_arrived + blk_finished == fibers.size()) { blk_arrived = 0; blk_gen++; }
	swapcontext(&f.ctx, &sched_ctx);
}
} // namespace

uint64_t x3emu_ballot(int p) { wave_rendezvous(p, 0); return waves[fibers[cur].wave].mask; }

uint32_t x3emu_shfl(uint32_t v, int src)
{
	wave_rendezvous(0, v);
	Wave &w = waves[fibers[cur].wave];
	if (src < 0 || src >= (int)w.size) src = (int)fibers[cur].lane;
	return w.res[src];
}

void x3emu_syncthreads()
{
	blk_arrived++;
	const unsigned long g = blk_gen;
	if (blk_arrived + blk_finished == fibers.size()) { blk_arrived = 0; blk_gen++; }
	else while (blk_gen == g) yield_to_sched();
}

void x3emu_launch(void (*fn)(void *), void *arg, dim3 grid, dim3 block)
{
	kern_fn = fn;
	kern_arg = arg;
	gridDim = grid;
	blockDim = block;
	const unsigned nthreads = block.x * block.y * block.z;
	const size_t stack_bytes = 256 * 1024;
	for (unsigned bz = 0; bz < grid.z; bz++)
	for (unsigned by = 0; by < grid.y; by++)
	for (unsigned bx = 0; bx < grid.x; bx++) {
		fibers.assign(nthreads, Fiber());
		waves.assign((nthreads + X3_WAVE - 1) / X3_WAVE, Wave());
		blk_arrived = blk_finished = 0;
		for (unsigned t = 0; t < nthreads; t++) {
			Fiber &f = fibers[t];
			f.stack.resize(stack_bytes);
			f.wave = t / X3_WAVE;
			f.lane = t % X3_WAVE;
			f.tid = x3emu_dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
			waves[f.wave].size++;
			getcontext(&f.ctx);
			f.ctx.uc_stack.ss_sp = f.stack.data();
			f.ctx.uc_stack.ss_size = stack_bytes;
			f.ctx.uc_link = &sched_ctx;
			makecontext(&f.ctx, (void (*)())fiber_main, 0);
		}
		for (auto &w : waves) for (unsigned l = 0; l < X3_WAVE; l++) w.here[l] = false;
		unsigned live = nthreads;
		unsigned long spins = 0;
		while (live) {
			live = 0;
			for (unsigned t = 0; t < nthreads; t++) {
				if (fibers[t].done) continue;
				live++;
				cur = (int)t;
				threadIdx = fibers[t].tid;
				blockIdx = x3emu_dim3(bx, by, bz);
				swapcontext(&sched_ctx, &fibers[t].ctx);
			}
			if (++spins > 2000000000ul) throw std::runtime_error("x3emu: kernel does not terminate (divergent rendezvous?)");
		}
	}
	fibers.clear();
	waves.clear();
	cur = -1;
}
