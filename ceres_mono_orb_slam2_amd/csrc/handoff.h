// Cross-workgroup hand-offs without cache maintenance (gfx942 / gfx950 only).
#pragma once
// Hand-offs between workgroups (the persistent factorisations of ba_solver.hip, the split-pair merge of orb_matcher.hip): the DATA another workgroup will read is written with
// agent-scope relaxed atomic stores (sc1: written through to the memory side, no L2 write-back) and read with agent-scope
// relaxed atomic loads; the flag is an agent-scope relaxed store / load.  What orders them is the hardware, not the language
// model: on gfx942 / gfx950 s_waitcnt vmcnt(0) returns only when this wave's stores have been acknowledged by the memory
// side, and a wave does not issue the loads behind a flag load it is still waiting for.  That is a property of THIS ISA
// (vmcnt does not cover stores from gfx10 on), hence the guard; the workgroup-scope fences are there for the COMPILER
// (s_waitcnt is IntrNoMem: without them stores could sink below it / loads rise above the flag load) and cost no cache
// maintenance at workgroup scope.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the relaxed agent-scope hand-offs of this library rely on gfx942 / gfx950 vmcnt semantics; port them to release / acquire first"
#endif
#define HANDOFF_DRAIN() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); } while (0)
#define HANDOFF_ACQUIRE() asm volatile("" ::: "memory")
