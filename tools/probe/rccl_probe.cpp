// probe: can two RCCL ranks share ONE GPU?  (fork before any GPU call; id through a pipe)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <sys/wait.h>
#include <chrono>
int main(int argc, char **argv)
{
  int world = argc > 1 ? atoi(argv[1]) : 2;
  int pipes[8][2];
  for (int r = 1; r < world; r++) if (pipe(pipes[r])) return 9;
  pid_t pids[8];
  int rank = 0;
  for (int r = 1; r < world; r++) { pid_t p = fork(); if (p == 0) { rank = r; break; } pids[r] = p; }
  ncclUniqueId id;
  if (rank == 0) { if (ncclGetUniqueId(&id) != ncclSuccess) { printf("getid failed\n"); return 1; } for (int r = 1; r < world; r++) write(pipes[r][1], &id, sizeof id); }
  else read(pipes[rank][0], &id, sizeof id);
  int ndev = 0; hipGetDeviceCount(&ndev);
  hipSetDevice(rank % ndev);
  ncclComm_t comm;
  ncclResult_t rc = ncclCommInitRank(&comm, world, id, rank);
  printf("rank %d: ncclCommInitRank -> %d (%s), ndev %d\n", rank, (int)rc, ncclGetErrorString(rc), ndev);
  if (rc == ncclSuccess) {
    double *din, *dout; hipStream_t st; hipStreamCreate(&st);
    hipMalloc(&din, 64 * 8); hipMalloc(&dout, 64 * 8 * world);
    double h[64]; for (int i = 0; i < 64; i++) h[i] = rank * 100 + i;
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < 100; k++) ncclAllGather(din, dout, 64, ncclDouble, comm, st);
      hipStreamSynchronize(st);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100;
      if (rank == 0) printf("allgather 512 B x %d ranks: %.1f us each (stream-queued)\n", world, us);
    }
    { auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < 100; k++) { ncclAllGather(din, dout, 64, ncclDouble, comm, st); hipStreamSynchronize(st); }
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100;
      if (rank == 0) printf("allgather + sync: %.1f us each\n", us); }
    double o[64 * 8]; hipMemcpy(o, dout, 64 * 8 * world, hipMemcpyDeviceToHost);
    printf("rank %d: got %g %g\n", rank, o[0], o[64 * (world - 1) + 1]);
    ncclCommDestroy(comm);
  }
  if (rank == 0) for (int r = 1; r < world; r++) { int s; waitpid(pids[r], &s, 0); }
  return rc == ncclSuccess ? 0 : 2;
}
