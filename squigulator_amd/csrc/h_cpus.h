// h_cpus.h -- how many CPUs the process may use (shared by the staging helpers and the BLOW5 writer's zlib threads; the
// BLOW5 writer is also compiled into the CPU backend of the test infrastructure, hence a header of its own)
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

// CPUs this process may actually use: the affinity mask, cut by the cgroup's CPU quota (v2: cpu.max, v1: cpu.cfs_quota_us /
// cpu.cfs_period_us) -- the GPU boxes show 256 hardware threads and allow 16
static int usable_cpus() {
    static const int n = [] {
        int cpus = (int)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::max(1, std::min(cpus, CPU_COUNT(&set)));
        double quota = 0.0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64]; double per = 0.0;
            if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / per;
            fclose(f);
        } else {
            double q = -1.0, per = 0.0;
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &q) != 1) q = -1.0; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &per) != 1) per = 0.0; fclose(g); }
            if (q > 0 && per > 0) quota = q / per;
        }
        if (quota >= 1.0) cpus = std::min(cpus, (int)quota);
        return std::max(1, cpus);
    }();
    return n;
}

