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
