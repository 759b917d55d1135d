/* boundary_consumer.c -- the trainer end of the wire protocol with nothing else in it: waits for a batch
 * (sem_w_<dev>_<pipe>), reads its counters from the server's host-visible mirror, releases the slot (sem_r_...).
 * Used by tools/server_throughput.py --consumer native to measure what the SERVER can hand over per second when
 * the consumer costs nothing (the Python consumer adds ~15 us of tensor wrapping per batch).
 *   gcc -O2 boundary_consumer.c -o boundary_consumer -lrt -lpthread
 *   boundary_consumer <namespace-suffix> <device> <hops> <skip> [epochs] [views]    -> one JSON line
 * `views` = 1: announce a trainer end that takes batches as views of the server's lane arena (this consumer touches no
 * device memory, so it never opens the arena: it measures the hand-over protocol of that mode).
 * Protocol: SS/engine/ipc_service.cu:28-31,181-192,283-291; TB/ipc_cuda_kernel.cu:75-106. */
#include <fcntl.h>
#include <semaphore.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define MAX_DEVICE 8
#define INTERBATCH_CON 2
#define MEMORY_USAGE 7
#define MAGIC 0x4C47494F
typedef struct {
    int32_t steps[3];
    char memHandle[MAX_DEVICE][INTERBATCH_CON][MEMORY_USAGE][64];
} shmStruct;
typedef struct {            /* "legionIPCext<suffix>": the server's host-visible counter mirror + direct-view hand-over (ipc_env.hip) */
    int32_t ext_magic;
    int32_t ext_version;
    int32_t server_state;
    int32_t ext_reserved;
    int32_t counters[MAX_DEVICE][INTERBATCH_CON][32];
    char arena[MAX_DEVICE][64];
    int64_t arena_bytes[MAX_DEVICE];
    int32_t trainer_direct[MAX_DEVICE];
    int32_t view_on[MAX_DEVICE][INTERBATCH_CON];
    int64_t view[MAX_DEVICE][INTERBATCH_CON][5];
} shmExt;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv)
{
    if (argc < 5 || argc > 7) { fprintf(stderr, "usage: boundary_consumer <suffix> <device> <hops> <skip> [epochs] [views]\n"); return 2; }
    const char* sfx = argv[1];
    const int dev = atoi(argv[2]), hops = atoi(argv[3]), skip = atoi(argv[4]), epochs = argc >= 6 ? atoi(argv[5]) : 1;
    const int views = argc >= 7 ? atoi(argv[6]) : 0;
    char name[128];
    snprintf(name, sizeof name, "simpleIPCshm%s", sfx);
    int fd = shm_open(name, O_RDWR, 0777);
    if (fd < 0) { perror("shm_open"); return 1; }
    volatile shmStruct* shm = (volatile shmStruct*)mmap(0, sizeof(shmStruct), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (shm == MAP_FAILED) { perror("mmap"); return 1; }
    snprintf(name, sizeof name, "legionIPCext%s", sfx);
    int efd = shm_open(name, O_RDWR, 0);
    if (efd < 0) { fprintf(stderr, "server does not publish the counter mirror\n"); return 1; }
    volatile shmExt* ext = (volatile shmExt*)mmap(0, sizeof(shmExt), PROT_READ | PROT_WRITE, MAP_SHARED, efd, 0);
    if (ext == MAP_FAILED || ext->ext_magic != MAGIC) { fprintf(stderr, "server does not publish the counter mirror\n"); return 1; }
    int took_views = 0;
    if (views && ext->ext_version >= 2 && ext->arena_bytes[dev] > 0) { ext->trainer_direct[dev] = 1; __sync_synchronize(); took_views = 1; }
    sem_t *sr[2], *sw[2];
    for (int i = 0; i < 2; i++) {
        snprintf(name, sizeof name, "sem_r_%d_%d%s", dev, i, sfx);
        sr[i] = sem_open(name, O_CREAT | O_RDWR, 0666, 0);
        snprintf(name, sizeof name, "sem_w_%d_%d%s", dev, i, sfx);
        sw[i] = sem_open(name, O_CREAT | O_RDWR, 0666, 0);
        if (sr[i] == SEM_FAILED || sw[i] == SEM_FAILED) { perror("sem_open"); return 1; }
        sem_post(sr[i]);                                   /* both pipe slots start free */
    }
    /* the reference's schedule: (train + valid) x epochs + test; timed up to the last training batch of the last epoch */
    const int per_epoch = shm->steps[0] + shm->steps[1];
    const int total = per_epoch * epochs + shm->steps[2], train = per_epoch * (epochs - 1) + shm->steps[0];
    long long edges = 0, nodes = 0;
    int timed = 0, pipe = 0;
    double t0 = 0, t1 = 0;
    for (int i = 0; i < total; i++) {
        sem_wait(sw[pipe]);
        if (i == skip) { t0 = now(); edges = nodes = 0; timed = 0; }
        if (i < train) {
            edges += ext->counters[dev][pipe][16 + 9 + hops];
            nodes += ext->counters[dev][pipe][9 + hops];
            timed++;
        }
        sem_post(sr[pipe]);
        pipe ^= 1;
        if (i == train - 1) t1 = now();
    }
    printf("{\"consumer\": \"native (protocol only)\", \"views\": %d, \"batches_per_sec\": %.1f, \"edges_per_sec\": %.1f, \"timed_batches\": %d, "
           "\"ms_per_batch\": %.6f, \"nodes_per_batch\": %.1f}\n", took_views, timed / (t1 - t0), edges / (t1 - t0), timed,
           (t1 - t0) / timed * 1e3, (double)nodes / timed);
    return 0;
}
