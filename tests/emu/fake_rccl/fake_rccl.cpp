// TEST INFRASTRUCTURE ONLY (not product code): the five RCCL entry points am355_shard.hip calls, implemented over files in a
// directory so that the sharded replay's collective can run between PROCESSES of the CPU emulation (its "device" pointers are host
// pointers). Selected with AM355_RCCL_LIB=tests/emu/libfake_rccl.so; the product opens the real librccl.so.1.
//   unique id  = the rendezvous directory (made by ncclGetUniqueId under $TMPDIR or /tmp)
//   all-gather = every rank writes <dir>/<seq>.<rank>.bin (via rename: complete or absent), then reads the others' in rank order
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

struct FakeComm { std::string dir; int rank, world; unsigned long seq; };
struct FakeId { char internal[128]; };

extern "C" int ncclGetUniqueId(FakeId* id) {
  const char* tmp = getenv("TMPDIR");
  char path[120];
  snprintf(path, sizeof path, "%s/am355_fake_rccl_XXXXXX", tmp && *tmp && strlen(tmp) < 80 ? tmp : "/tmp");
  if (!mkdtemp(path)) return 2;
  memset(id->internal, 0, sizeof id->internal);
  strncpy(id->internal, path, sizeof id->internal - 1);
  return 0;
}
extern "C" int ncclCommInitRank(void** comm, int nranks, FakeId id, int rank) {
  id.internal[127] = 0;
  struct stat sb;
  if (stat(id.internal, &sb) != 0) return 2;
  *comm = new FakeComm{id.internal, rank, nranks, 0};
  return 0;
}
extern "C" int ncclAllGather(const void* send, void* recv, size_t count, int datatype, void* comm_, void* /*stream*/) {
  if (datatype != 1) return 4;  // (ncclUint8 is all the engine sends)
  FakeComm* c = (FakeComm*)comm_;
  const unsigned long seq = c->seq++;
  char tmp[256], fin[256];
  snprintf(tmp, sizeof tmp, "%s/%lu.%d.tmp", c->dir.c_str(), seq, c->rank);
  snprintf(fin, sizeof fin, "%s/%lu.%d.bin", c->dir.c_str(), seq, c->rank);
  FILE* f = fopen(tmp, "wb");
  if (!f) return 2;
  if (count && fwrite(send, 1, count, f) != count) { fclose(f); return 2; }
  fclose(f);
  if (rename(tmp, fin) != 0) return 2;
  for (int r = 0; r < c->world; r++) {
    snprintf(fin, sizeof fin, "%s/%lu.%d.bin", c->dir.c_str(), seq, r);
    FILE* g = nullptr;
    for (int spins = 0; !(g = fopen(fin, "rb")); spins++) {
      if (spins > 600000) return 6;  // (~60 s: a peer that never arrives must fail the test, not hang it)
      struct timespec ts = {0, 100000};
      nanosleep(&ts, nullptr);
    }
    size_t got = count ? fread((uint8_t*)recv + (size_t)r * count, 1, count, g) : 0;
    fclose(g);
    if (got != count) return 2;
  }
  return 0;
}
extern "C" int ncclCommDestroy(void* comm_) {
  FakeComm* c = (FakeComm*)comm_;
  if (c->rank == 0) {  // (leaves the files of unfinished peers alone: rank 0 removes what it can at the end)
    std::string cmd = "rm -rf '" + c->dir + "'";
    if (c->dir.find("am355_fake_rccl_") != std::string::npos) (void)!system(cmd.c_str());
  }
  delete c;
  return 0;
}
extern "C" const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : rc == 6 ? "fake rccl: a peer did not arrive" : "fake rccl: file I/O failed"; }
