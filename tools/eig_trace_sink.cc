// Sink of the ORACLE_EIG_TRACE events of oracle/linalg.h (tools/sim_roots_lanes.py): packs (kind, il, imm, iu) into one int per event.
#include <vector>
#include <cstdint>
static thread_local std::vector<int32_t> g_trace;
extern "C" void oracle_eig_trace(int kind, int il, int imm, int iu) { g_trace.push_back(kind | (il << 4) | (imm << 8) | (iu << 12)); }
extern "C" int oracle_eig_trace_take(int32_t* out, int cap) { int n = (int)g_trace.size(); for (int i = 0; i < n && i < cap; ++i) out[i] = g_trace[i]; g_trace.clear(); return n; }
