/* rccl_stub.h -- X3_EMU builds only (tests): the six librccl entry points x3h_compress_container_rccl uses, modelled on host memory so that
 * the multi-rank partition / pack / offset / header code of api.hip runs with 2..8 "devices" on a machine that has none.
 *   ncclSend / ncclRecv inside ncclGroupStart..ncclGroupEnd are RECORDED; ncclGroupEnd pairs every receive (at rank r, from peer p) with the
 *   oldest unmatched send (from rank p, to peer r), insists on equal byte counts -- a real RCCL would hang or corrupt on a mismatch -- and
 *   executes the pair as one memcpy.  A send or receive left unmatched, a count mismatch, an operation outside a group or a peer out of
 *   range fails the group (ncclInvalidUsage), which the caller reports as X3H_E_RCCL.
 * Counters of the last completed group are exported (x3emu_rccl_last_group) so that a test can assert "ONE group, nd - 1 pairs, exact sizes". */
#ifndef X3_RCCL_STUB_H
#define X3_RCCL_STUB_H
#include <stdint.h>
#include <string.h>
#include <vector>

struct ncclComm { int rank, nranks; };
namespace x3emu_rccl {
struct Op { bool send; int at, peer; void *buf; size_t count; bool done; };
static std::vector<Op> g_ops;
static int g_depth = 0;
static bool g_bad = false;
static uint64_t g_groups = 0, g_last_pairs = 0, g_last_bytes = 0;
enum { Success = 0, InvalidUsage = 5 };

static int CommInitAll(ncclComm **comms, int n, const int *)
{
	for (int i = 0; i < n; i++) { comms[i] = new ncclComm(); comms[i]->rank = i; comms[i]->nranks = n; }
	return Success;
}
static int CommDestroy(ncclComm *c) { delete c; return Success; }
static int GroupStart() { if (g_depth++ == 0) { g_ops.clear(); g_bad = false; } return Success; }
static int record(bool send, void *buf, size_t count, int dtype, int peer, ncclComm *comm)
{
	if (g_depth <= 0 || !comm || peer < 0 || peer >= comm->nranks || dtype != 1 /* ncclUint8 */ || (!buf && count)) { g_bad = true; return InvalidUsage; }
	g_ops.push_back(Op{ send, comm->rank, peer, buf, count, false });
	return Success;
}
static int Send(const void *buf, size_t count, int dtype, int peer, ncclComm *comm, void *) { return record(true, (void *)buf, count, dtype, peer, comm); }
static int Recv(void *buf, size_t count, int dtype, int peer, ncclComm *comm, void *) { return record(false, buf, count, dtype, peer, comm); }
static int GroupEnd()
{
	if (g_depth <= 0) return InvalidUsage;
	if (--g_depth) return Success;
	uint64_t pairs = 0, bytes = 0;
	bool ok = !g_bad;
	for (Op &r : g_ops) {
		if (r.send || !ok) continue;
		Op *s = nullptr;
		for (Op &c : g_ops) if (c.send && !c.done && c.at == r.peer && c.peer == r.at) { s = &c; break; }
		if (!s || s->count != r.count) { ok = false; break; }
		memmove(r.buf, s->buf, r.count);
		s->done = r.done = true;
		pairs++; bytes += r.count;
	}
	for (const Op &o : g_ops) if (!o.done) ok = false; /* a send nobody receives (or the reverse) would hang a real communicator */
	g_ops.clear();
	if (!ok) return InvalidUsage;
	g_groups++; g_last_pairs = pairs; g_last_bytes = bytes;
	return Success;
}
} // namespace x3emu_rccl

/* (groups completed so far, pairs and bytes of the last one) */
extern "C" void x3emu_rccl_last_group(uint64_t *groups, uint64_t *pairs, uint64_t *bytes)
{
	*groups = x3emu_rccl::g_groups; *pairs = x3emu_rccl::g_last_pairs; *bytes = x3emu_rccl::g_last_bytes;
}
#endif
