/*
 * fake_rccl.c — librccl.so.1 for the kernel emulation (TEST INFRASTRUCTURE): the RCCL entry points crh_frames_reduce() / crh_frames_gather() bind
 * (csrc/cray_hip.hip: ncclCommInitAll, ncclGroupStart / ncclGroupEnd, ncclReduce, ncclSend / ncclRecv, ncclGetErrorString) on "devices" that are all this
 * process's heap. A group of ncclReduce calls — one per rank, float32, sum, in place — is carried out at ncclGroupEnd: root's buffer
 * becomes the element-wise sum of all ranks' buffers, added in rank order. Found through LD_LIBRARY_PATH by the drop-in program when the
 * CPU tier runs it with several emulated devices; it checks what the product promises RCCL (same count / type / op / root on every rank,
 * one call per communicator) and fails loudly otherwise.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct fake_comm { int rank, size; } fake_comm;
enum { MAX_RANKS = 16 };
static struct { const float *send; float *recv; size_t count; int root; fake_comm *comm; } g_call[MAX_RANKS];
static int g_calls, g_in_group;

int ncclCommInitAll(fake_comm **comms, int n, const int *devices) {
	(void)devices;
	if (!comms || n < 1 || n > MAX_RANKS) return 4;       /* ncclInvalidArgument */
	for (int i = 0; i < n; ++i) {
		comms[i] = malloc(sizeof(fake_comm));
		if (!comms[i]) return 1;
		comms[i]->rank = i; comms[i]->size = n;
	}
	return 0;
}
static int finishP2p(void);
static int g_p2ps_reset(void);
int ncclGroupStart(void) { g_in_group = 1; g_calls = 0; return g_p2ps_reset(); }
int ncclReduce(const void *send, void *recv, size_t count, int datatype, int op, int root, fake_comm *comm, void *stream) {
	(void)stream;
	if (!g_in_group || !comm || datatype != 7 /* ncclFloat32 */ || op != 0 /* ncclSum */ || g_calls >= MAX_RANKS) return 4;
	g_call[g_calls].send = send; g_call[g_calls].recv = recv; g_call[g_calls].count = count; g_call[g_calls].root = root; g_call[g_calls].comm = comm;
	++g_calls;
	return 0;
}
int ncclGroupEnd(void) {
	g_in_group = 0;
	if (g_calls == 0) return finishP2p();
	const int n = g_call[0].comm->size, root = g_call[0].root;
	if (g_calls != n) { fprintf(stderr, "fake rccl: %d ncclReduce calls for a communicator of %d ranks\n", g_calls, n); return 5; }
	float *out = NULL;
	unsigned seen = 0;
	for (int i = 0; i < n; ++i) {
		if (g_call[i].count != g_call[0].count || g_call[i].root != root || g_call[i].comm->size != n) return 4;
		seen |= 1u << g_call[i].comm->rank;
		if (g_call[i].comm->rank == root) out = g_call[i].recv;
	}
	if (seen != (1u << n) - 1u || !out) { fprintf(stderr, "fake rccl: ranks missing from the group\n"); return 5; }
	float *acc = calloc(g_call[0].count ? g_call[0].count : 1, sizeof(float));
	if (!acc) return 1;
	for (int r = 0; r < n; ++r)                                   /* rank order */
		for (int i = 0; i < n; ++i)
			if (g_call[i].comm->rank == r)
				for (size_t k = 0; k < g_call[0].count; ++k) acc[k] += g_call[i].send[k];
	memcpy(out, acc, g_call[0].count * sizeof(float));
	free(acc);
	g_calls = 0;
	return 0;
}
/* ncclSend / ncclRecv (crh_frames_gather): recorded inside a group, matched at ncclGroupEnd — every send needs a receive of the same count
 * posted by its peer with the sender's rank, and the other way round. */
static struct { const void *send; void *recv; size_t count; int peer; fake_comm *comm; } g_p2p[4 * MAX_RANKS];
static int g_p2ps;
int ncclSend(const void *send, size_t count, int datatype, int peer, fake_comm *comm, void *stream) {
	(void)stream;
	if (!g_in_group || !comm || datatype != 7 || g_p2ps >= 4 * MAX_RANKS || peer < 0 || peer >= comm->size || peer == comm->rank) return 4;
	g_p2p[g_p2ps].send = send; g_p2p[g_p2ps].recv = NULL; g_p2p[g_p2ps].count = count; g_p2p[g_p2ps].peer = peer; g_p2p[g_p2ps].comm = comm;
	++g_p2ps;
	return 0;
}
int ncclRecv(void *recv, size_t count, int datatype, int peer, fake_comm *comm, void *stream) {
	(void)stream;
	if (!g_in_group || !comm || datatype != 7 || g_p2ps >= 4 * MAX_RANKS || peer < 0 || peer >= comm->size || peer == comm->rank) return 4;
	g_p2p[g_p2ps].send = NULL; g_p2p[g_p2ps].recv = recv; g_p2p[g_p2ps].count = count; g_p2p[g_p2ps].peer = peer; g_p2p[g_p2ps].comm = comm;
	++g_p2ps;
	return 0;
}
static int g_p2ps_reset(void) { g_p2ps = 0; return 0; }
static int finishP2p(void) {
	int matched = 0;
	for (int i = 0; i < g_p2ps; ++i) {
		if (!g_p2p[i].send) continue;
		int found = 0;
		for (int j = 0; j < g_p2ps && !found; ++j) {
			if (!g_p2p[j].recv || g_p2p[j].comm->rank != g_p2p[i].peer || g_p2p[j].peer != g_p2p[i].comm->rank) continue;
			if (g_p2p[j].count != g_p2p[i].count) { fprintf(stderr, "fake rccl: send of %zu floats meets a receive of %zu\n", g_p2p[i].count, g_p2p[j].count); return 5; }
			memcpy(g_p2p[j].recv, g_p2p[i].send, g_p2p[i].count * sizeof(float));
			g_p2p[j].recv = NULL;
			found = 1; matched += 2;
		}
		if (!found) { fprintf(stderr, "fake rccl: rank %d sends to %d, which posted no receive\n", g_p2p[i].comm->rank, g_p2p[i].peer); return 5; }
	}
	if (matched != g_p2ps) { fprintf(stderr, "fake rccl: %d receives without a send\n", g_p2ps - matched); return 5; }
	g_p2ps = 0;
	return 0;
}
const char *ncclGetErrorString(int code) { return code == 0 ? "no error" : code == 4 ? "invalid argument" : code == 5 ? "invalid usage" : "error"; }
