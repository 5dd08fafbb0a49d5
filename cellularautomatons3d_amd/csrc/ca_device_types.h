// Plain structs shared by host code and device code — including the device sources compiled at run time by
// hiprtc (ca_jit.cpp), so: no standard headers beyond fixed-width integers, no host-only types.
#pragma once

#ifdef __HIPCC_RTC__
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned char uint8_t;
#else
#include <stdint.h>
#endif

namespace ca3d
{

// ---------------------------------------------------------------------------------------------- rule programs
//
// A born/survive LUT slice (27 slots of one rule-set) is a boolean function of the bit-sliced neighbour count.
// The host compiles it (Quine-McCluskey with the unreachable counts as don't-cares) into a short OR-of-cubes
// program which the kernels interpret with wave-uniform control flow: the LUT never reaches the GPU.
constexpr int kMaxCubes = 16;

struct RuleProg
{
	uint32_t n;      // cubes used
	uint32_t invert; // 0 or 0xFFFFFFFF: the cubes cover the complement
	// bits 0-4: care mask over count planes, bits 8-12: required plane value where cared
	uint32_t cubes[kMaxCubes];
};

struct RuleSetProg
{
	RuleProg born, survive;
};

struct PackedRuleArgs
{
	RuleSetProg set[3]; // main, edges, corners
};

enum MainKind : int
{
	MAIN_VN = 0,
	MAIN_VN2D = 1,
	MAIN_MOORE = 2,
	MAIN_MOORE2D = 3,
	MAIN_EDGES = 4,
	MAIN_CORNERS = 5,
	MAIN_GENERIC = 6
};

// A step over output planes [lo, hi) (and optionally [lo2, hi2)) of an array of `nplanes` z-planes; plane j holds global z = zbase + j
// (mod G). Full grid: zbase 0, nplanes G, wrap_full 1. See oracle/ca_oracle.c for the same convention.
struct PlaneRange
{
	uint32_t G;
	uint32_t nplanes;
	int32_t zbase;
	uint32_t lo, hi;
	uint32_t wrap_full;
	uint32_t lo2 = 0, hi2 = 0; // optional second output range (the two edge zones of a slab batch in one launch)
};

} // namespace ca3d
