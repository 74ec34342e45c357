#pragma once

// The lanes of a wavefront execute in lock step, so program order alone orders "every lane reads X, then one lane
// overwrites X" and "one lane writes X, then every lane reads X" within a wavefront.  ND_LOCKSTEP() marks the places the
// kernels rely on that.  It compiles to nothing; the lane-by-lane CPU interpreter of the kernel tests (tests/simt, which
// defines SIMT_EMULATION) runs lanes one after the other and turns the mark into a rendezvous of the wavefront.
#ifdef SIMT_EMULATION
#define ND_LOCKSTEP() __builtin_amdgcn_wave_barrier()
#else
#define ND_LOCKSTEP() ((void)0)
#endif

// ND_LOADED(x): the value of x is in its register from here on.  The compiler waits for a load at the first use of its result; when
// the load sits in a branch of a loop and the use behind the branch, that wait (s_waitcnt vmcnt(0): every store and atomic issued
// before it, too) is paid by every round of the loop.  The mark is a use inside the branch.
#ifdef SIMT_EMULATION
#define ND_LOADED(x) ((void)(x))
#else
#define ND_LOADED(x) asm volatile("" : "+v"(x))
#endif
