// The CPUs of one NUMA node (/sys/devices/system/node/nodeN/cpulist), cut down to what the process may use: where the host threads that feed a device
// are bound (lamd_served's engine threads, lamd_multi's workers) once lamd_device_numa_node() has said which node the device hangs on.
#pragma once
#include <sched.h>
#include <stdio.h>

static inline bool lamd_node_cpus(int node, cpu_set_t *out) {
  char path[96];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return false;
  CPU_ZERO(out);
  int a, b, n = 0;
  for (;;) {
    if (fscanf(f, "%d", &a) != 1) break;
    b = a;
    int c = fgetc(f);
    if (c == '-') {
      if (fscanf(f, "%d", &b) != 1) break;
      c = fgetc(f);
    }
    for (int i = a; i <= b && i < CPU_SETSIZE; i++, n++) CPU_SET(i, out);
    if (c != ',') break;
  }
  fclose(f);
  cpu_set_t mine;
  if (sched_getaffinity(0, sizeof mine, &mine) == 0) CPU_AND(out, out, &mine);  // never outside what the process may use
  return n > 0 && CPU_COUNT(out) > 0;
}
