/* LD_PRELOAD aid: print a backtrace to stderr (and to $SEGV_BT_FILE if set) when the process receives SIGSEGV / SIGABRT / SIGBUS. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig)
{
  void* bt[64];
  int n = backtrace(bt, 64);
  const char* f = getenv("SEGV_BT_FILE");
  int fd = f ? open(f, O_WRONLY | O_CREAT | O_APPEND, 0644) : 2;
  if(fd < 0) fd = 2;
  const char* msg = "\n=== signal caught, backtrace ===\n";
  if(write(fd, msg, strlen(msg)) < 0) {}
  backtrace_symbols_fd(bt, n, fd);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void init(void)
{
  signal(SIGSEGV, handler);
  signal(SIGBUS, handler);
  signal(SIGABRT, handler);
}
