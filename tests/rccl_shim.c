/* TEST INFRASTRUCTURE -- a stand-in for librccl between OS processes that share ONE GPU.
 *
 * The gpurun box has a single MI355X whose compute partitioning cannot be changed from inside the
 * container (profiles/r05_partition_probe.txt), so RCCL itself never sees more than one rank there.
 * What can still be executed with several REAL ranks is everything of the library around the
 * collectives: csrc/capi_colpart.inc's one-process-per-GPU loop (ncclCommInitRank, the blind enqueue,
 * exchange A = all-gather of the pricing winners, exchange B = int64 SUM all-reduce / rooted
 * broadcast of the entering column, termination and the read-back on every rank).  The library binds
 * RCCL at run time by NAME (dl_iterate_phdr / dlopen of "librccl.so*", dlsym of nine entry points);
 * this file, built as librccl.so.1 by tests/test_colpart_rccl_shim.py and loaded by the worker
 * processes before the library, provides those nine with the collectives' semantics: stream-ordered
 * (the call waits for the stream, stages through the host and returns when the result is in the
 * receive buffer) over TCP on 127.0.0.1, star-shaped through rank 0.  Nothing in the product links or
 * loads this file. */
#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

struct ncclComm {
    int rank, nranks;
    int fds[64];              /* rank 0: socket of every peer; others: fds[0] = socket to rank 0 */
    char *stage;              /* nranks x max message */
    size_t stage_cap;
};

static int64_t g_calls[4];    /* all-gathers, all-reduces, broadcasts, bytes exchanged */
static int g_listen_fd = -1, g_listen_port = 0;

void rccl_shim_stats(int64_t *out4) { memcpy(out4, g_calls, sizeof g_calls); }

static int send_all(int fd, const void *buf, size_t n)
{
    const char *p = (const char *)buf;
    while (n) {
        ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
        if (k < 0) { if (errno == EINTR) continue; return -1; }
        p += k; n -= (size_t)k;
    }
    return 0;
}
static int recv_all(int fd, void *buf, size_t n)
{
    char *p = (char *)buf;
    while (n) {
        ssize_t k = recv(fd, p, n, 0);
        if (k == 0) return -1;
        if (k < 0) { if (errno == EINTR) continue; return -1; }
        p += k; n -= (size_t)k;
    }
    return 0;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rccl shim: transport or HIP failure"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    struct sockaddr_in a;
    socklen_t len = sizeof a;
    int one = 1;
    memset(id, 0, sizeof *id);
    g_listen_fd = socket(AF_INET, SOCK_STREAM, 0);
    if (g_listen_fd < 0) return ncclSystemError;
    setsockopt(g_listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    memset(&a, 0, sizeof a);
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;
    if (bind(g_listen_fd, (struct sockaddr *)&a, sizeof a) != 0 || listen(g_listen_fd, 64) != 0) return ncclSystemError;
    if (getsockname(g_listen_fd, (struct sockaddr *)&a, &len) != 0) return ncclSystemError;
    g_listen_port = ntohs(a.sin_port);
    memcpy(id->internal, "MI355XSHIM", 10);
    memcpy(id->internal + 16, &g_listen_port, sizeof g_listen_port);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    struct ncclComm *c;
    int port = 0, one = 1;
    if (memcmp(id.internal, "MI355XSHIM", 10) != 0 || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    memcpy(&port, id.internal + 16, sizeof port);
    c = (struct ncclComm *)calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    c->rank = rank; c->nranks = nranks;
    for (int i = 0; i < 64; ++i) c->fds[i] = -1;
    if (rank == 0) {
        if (port != g_listen_port || g_listen_fd < 0) { free(c); return ncclInvalidArgument; }
        for (int k = 1; k < nranks; ++k) {
            int fd = accept(g_listen_fd, NULL, NULL), r = -1;
            if (fd < 0 || recv_all(fd, &r, sizeof r) != 0 || r < 1 || r >= nranks || c->fds[r] != -1) { free(c); return ncclSystemError; }
            setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
            c->fds[r] = fd;
        }
        close(g_listen_fd); g_listen_fd = -1;
    } else {
        struct sockaddr_in a;
        int fd = -1;
        memset(&a, 0, sizeof a);
        a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = htons((uint16_t)port);
        for (int tries = 0; tries < 6000; ++tries) {           /* rank 0 may not be listening yet */
            struct timespec ts = {0, 10 * 1000 * 1000};
            fd = socket(AF_INET, SOCK_STREAM, 0);
            if (fd >= 0 && connect(fd, (struct sockaddr *)&a, sizeof a) == 0) break;
            if (fd >= 0) close(fd);
            fd = -1;
            nanosleep(&ts, NULL);
        }
        if (fd < 0 || send_all(fd, &rank, sizeof rank) != 0) { free(c); return ncclSystemError; }
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        c->fds[0] = fd;
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    (void)comms; (void)ndev; (void)devlist;
    return ncclInvalidUsage;      /* one process, several GPUs: not what this stand-in is for */
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    for (int i = 0; i < 64; ++i) if (c->fds[i] >= 0) close(c->fds[i]);
    free(c->stage);
    free(c);
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c) { return ncclCommDestroy(c); }

/* every rank's `len` bytes -> all of them, in rank order, on every rank (c->stage) */
static int exchange(struct ncclComm *c, const void *mine, size_t len)
{
    const size_t total = len * (size_t)c->nranks;
    if (c->stage_cap < total) {
        free(c->stage);
        c->stage = (char *)malloc(total);
        c->stage_cap = c->stage ? total : 0;
        if (!c->stage) return -1;
    }
    memcpy(c->stage + len * (size_t)c->rank, mine, len);
    if (c->rank == 0) {
        for (int r = 1; r < c->nranks; ++r) if (recv_all(c->fds[r], c->stage + len * (size_t)r, len) != 0) return -1;
        for (int r = 1; r < c->nranks; ++r) if (send_all(c->fds[r], c->stage, total) != 0) return -1;
    } else {
        if (send_all(c->fds[0], mine, len) != 0 || recv_all(c->fds[0], c->stage, total) != 0) return -1;
    }
    g_calls[3] += (int64_t)total;
    return 0;
}

static size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

/* device buffer -> host copy, after everything enqueued on `stream` before the call */
static char *fetch(const void *dev, size_t bytes, hipStream_t stream)
{
    char *h = (char *)malloc(bytes ? bytes : 1);
    if (!h) return NULL;
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(h); return NULL; }
    return h;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype,
                           ncclComm_t c, hipStream_t stream)
{
    const size_t len = sendcount * type_size(datatype);
    char *mine;
    if (!c || !len) return ncclInvalidArgument;
    if (!(mine = fetch(sendbuff, len, stream))) return ncclUnhandledCudaError;
    if (exchange(c, mine, len) != 0) { free(mine); return ncclSystemError; }
    free(mine);
    if (hipMemcpy(recvbuff, c->stage, len * (size_t)c->nranks, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    g_calls[0]++;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t c, hipStream_t stream)
{
    const size_t len = count * 8;
    int64_t *mine, *acc;
    if (!c || !count || datatype != ncclInt64 || op != ncclSum) return ncclInvalidArgument;   /* the one form the library uses */
    if (!(mine = (int64_t *)fetch(sendbuff, len, stream))) return ncclUnhandledCudaError;
    if (exchange(c, mine, len) != 0) { free(mine); return ncclSystemError; }
    acc = mine;
    memset(acc, 0, len);
    for (int r = 0; r < c->nranks; ++r) {
        const int64_t *src = (const int64_t *)(c->stage + len * (size_t)r);
        for (size_t i = 0; i < count; ++i) acc[i] = (int64_t)((uint64_t)acc[i] + (uint64_t)src[i]);
    }
    if (hipMemcpy(recvbuff, acc, len, hipMemcpyHostToDevice) != hipSuccess) { free(mine); return ncclUnhandledCudaError; }
    free(mine);
    g_calls[1]++;
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root,
                           ncclComm_t c, hipStream_t stream)
{
    const size_t len = count * type_size(datatype);
    char *mine;
    if (!c || !len || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (!(mine = fetch(sendbuff, len, stream))) return ncclUnhandledCudaError;
    if (exchange(c, mine, len) != 0) { free(mine); return ncclSystemError; }
    free(mine);
    if (hipMemcpy(recvbuff, c->stage + len * (size_t)root, len, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    g_calls[2]++;
    return ncclSuccess;
}
