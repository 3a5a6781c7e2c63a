// gte_hot_nt.hip — the hot step kernel with NON-TEMPORAL observation stores, in its own
// translation unit like gte_hot.hip (same source, one macro).  Once the observation buffer
// no longer fits the 256 MB Infinity Cache the stores are a pure stream, and `nt` measures
// 35-60 % faster than sc1 there (131 072 envs: 87 us vs 118 us; 262 144: 162 us vs 261 us),
// while sc1 wins below (65 536 envs: 42.5 us vs 45.8 us) — profiles/r01_tune_store_policy.log.
#define GTE_HOT_NT 1
#define GTE_HOT_NAME(x) x##_nt
#include "gte_hot.hip"
