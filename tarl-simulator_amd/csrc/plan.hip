// plan.hip — static per-graph plan (host-built CSC/CSR, uploaded once) + error plumbing.
//
// Replaces what the reference recomputes every step on static topology: GraphDistribution.__init__'s
// sort / argsort / unique / boundary masks (src/reinforcement_learning.py:21-35) and PyG's edge gathers
// (src/direction_mpnn.py:230, src/response_mpnn.py:40).
#include <stdarg.h>
#include <string.h>

#include <map>
#include <vector>

#include "tarl_common.h"

static thread_local char g_err[512] = "";

void tarl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tarl_last_error(void) { return g_err; }
extern "C" int tarl_abi_version(void) { return TARL_ABI_VERSION; }
// the experiment flags this library was built with (`make variant EXPFLAGS=...`); empty for the product build
#ifndef TARL_BUILD_FLAGS
#define TARL_BUILD_FLAGS ""
#endif
extern "C" const char* tarl_build_flags(void) { return TARL_BUILD_FLAGS; }

static int upload(int32_t** dptr, const std::vector<int32_t>& h) {
  const size_t bytes = (h.empty() ? 1 : h.size()) * sizeof(int32_t);
  TARL_CHECK_HIP(hipMalloc((void**)dptr, bytes));
  if (!h.empty()) TARL_CHECK_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return TARL_OK;
}

extern "C" int tarl_plan_create(const int64_t* ei, int64_t E, int64_t N, const int64_t* src_order,
                                tarl_plan** out) {
  TARL_REQUIRE(out != nullptr, "out is null");
  *out = nullptr;
  TARL_REQUIRE(E >= 0 && N >= 0, "negative size");
  TARL_REQUIRE(E == 0 || ei != nullptr, "edge_index_host is null");
  TARL_REQUIRE(E < (int64_t)1 << 31 && N < (int64_t)1 << 31, "graph too large for the int32 plan");
  const int64_t* s = ei;
  const int64_t* d = ei + E;
  for (int64_t e = 0; e < E; ++e) {
    if (s[e] < 0 || s[e] >= N || d[e] < 0 || d[e] >= N) {
      tarl_set_error("tarl_plan_create: edge %lld = (%lld -> %lld) out of range for %lld nodes", (long long)e,
                     (long long)s[e], (long long)d[e], (long long)N);
      return TARL_ERR_INVALID;
    }
  }
  std::vector<int32_t> in_ptr(N + 1, 0), out_ptr(N + 1, 0), in_src(E), in_eid(E), out_dst(E), out_eid(E), src32(E),
      dst32(E), group_of_node(N, -1), node_of_group;
  for (int64_t e = 0; e < E; ++e) {
    ++in_ptr[d[e] + 1];
    ++out_ptr[s[e] + 1];
    src32[e] = (int32_t)s[e];
    dst32[e] = (int32_t)d[e];
  }
  int32_t max_in = 0, max_out = 0;
  for (int64_t n = 0; n < N; ++n) {
    if (in_ptr[n + 1] > max_in) max_in = in_ptr[n + 1];
    if (out_ptr[n + 1] > max_out) max_out = out_ptr[n + 1];
    if (out_ptr[n + 1] > 0) {
      group_of_node[n] = (int32_t)node_of_group.size();
      node_of_group.push_back((int32_t)n);
    }
    in_ptr[n + 1] += in_ptr[n];
    out_ptr[n + 1] += out_ptr[n];
  }
  {  // counting sort, stable in the original edge id
    std::vector<int32_t> cur(in_ptr.begin(), in_ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) {
      const int32_t k = cur[d[e]]++;
      in_src[k] = (int32_t)s[e];
      in_eid[k] = (int32_t)e;
    }
  }
  bool src_sorted = true, dst_sorted = true;
  if (src_order == nullptr) {
    std::vector<int32_t> cur(out_ptr.begin(), out_ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) {
      const int32_t k = cur[s[e]]++;
      out_dst[k] = (int32_t)d[e];
      out_eid[k] = (int32_t)e;
    }
  } else {  // caller-supplied sorted order (must be a permutation that sorts the sources)
    std::vector<char> seen(E, 0);
    for (int64_t k = 0; k < E; ++k) {
      const int64_t e = src_order[k];
      if (e < 0 || e >= E || seen[e]) {
        tarl_set_error("tarl_plan_create: src_order is not a permutation (position %lld)", (long long)k);
        return TARL_ERR_INVALID;
      }
      seen[e] = 1;
      if (k > 0 && s[src_order[k - 1]] > s[e]) {
        tarl_set_error("tarl_plan_create: src_order does not sort edge_index[0] (position %lld)", (long long)k);
        return TARL_ERR_INVALID;
      }
      out_dst[k] = (int32_t)d[e];
      out_eid[k] = (int32_t)e;
    }
  }
  for (int64_t k = 0; k < E; ++k) {
    if (out_eid[k] != k) src_sorted = false;
    if (in_eid[k] != k) dst_sorted = false;
  }

  tarl_plan* p = new (std::nothrow) tarl_plan();
  if (!p) {
    tarl_set_error("tarl_plan_create: out of host memory");
    return TARL_ERR_NOMEM;
  }
  memset(p, 0, sizeof(*p));
  p->N = N;
  p->E = E;
  p->G = (int64_t)node_of_group.size();
  p->max_in = max_in;
  p->max_out = max_out;
  p->src_sorted = src_sorted ? 1 : 0;
  p->dst_sorted = dst_sorted ? 1 : 0;
  {
    bool sib = N >= 4 && N % 4 == 0;
    for (int64_t c = 0; sib && c < N; c += 4) {
      const int32_t a0 = in_ptr[c], deg = in_ptr[c + 1] - a0;
      for (int64_t r = 1; sib && r < 4; ++r) {
        const int32_t ar = in_ptr[c + r];
        sib = in_ptr[c + r + 1] - ar == deg;
        for (int32_t q = 0; sib && q < deg && q < 4; ++q) sib = in_src[ar + q] == in_src[a0 + q];
      }
    }
    p->siblings4 = sib ? 1 : 0;
  }
  // Row chunks of the row pass: rows grouped by their ordered out-edge target list, four per chunk (a group's remainder
  // makes a partial chunk, padded with -1). Groups in order of their first row, rows ascending inside a group. Worth using
  // when the table is not much longer than N / 4 chunks, i.e. when most rows share their targets with three others.
  std::vector<int32_t> row_chunks;
  {
    std::map<std::vector<int32_t>, int32_t> group_of_list;
    std::vector<std::vector<int32_t>> groups;
    for (int64_t n = 0; n < N; ++n) {
      std::vector<int32_t> key(out_dst.begin() + out_ptr[n], out_dst.begin() + out_ptr[n + 1]);
      auto it = group_of_list.find(key);
      if (it == group_of_list.end()) {
        group_of_list.emplace(std::move(key), (int32_t)groups.size());
        groups.emplace_back(1, (int32_t)n);
      } else {
        groups[it->second].push_back((int32_t)n);
      }
    }
    for (const auto& g : groups)
      for (size_t k = 0; k < g.size(); k += 4) {
        for (size_t r = 0; r < 4; ++r) row_chunks.push_back(k + r < g.size() ? g[k + r] : -1);
        const int32_t n0 = g[k];
        for (int32_t q = 0; q < 4; ++q)   // beyond the out-degree: the row itself (valid, ignored by the readers), as NodeRec::out4
          row_chunks.push_back(q < out_ptr[n0 + 1] - out_ptr[n0] ? out_dst[out_ptr[n0] + q] : n0);
      }
    p->num_row_chunks = (int64_t)(row_chunks.size() / 8);
    p->row_siblings = (N >= 8 && p->num_row_chunks * 4 <= N + N / 4) ? 1 : 0;
  }
  int rc;
#define UP(field, vec)                       \
  if ((rc = upload(&p->field, vec)) != TARL_OK) { \
    tarl_plan_destroy(p);                    \
    return rc;                               \
  }
  UP(row_chunks, row_chunks)
  UP(in_ptr, in_ptr)
  UP(in_src, in_src)
  UP(in_eid, in_eid)
  UP(out_ptr, out_ptr)
  UP(out_dst, out_dst)
  UP(out_eid, out_eid)
  UP(src, src32)
  UP(dst, dst32)
  UP(group_of_node, group_of_node)
  UP(node_of_group, node_of_group)
#undef UP
  *out = p;
  return TARL_OK;
}

extern "C" void tarl_plan_destroy(tarl_plan* p) {
  if (!p) return;
  int32_t* arrs[] = {p->row_chunks, p->in_ptr, p->in_src, p->in_eid, p->out_ptr, p->out_dst, p->out_eid, p->src, p->dst,
                     p->group_of_node, p->node_of_group};
  for (int32_t* a : arrs)
    if (a) (void)hipFree(a);
  delete p;
}

extern "C" int tarl_plan_geometry(const tarl_plan* p, int64_t* info) {
  TARL_REQUIRE(p != nullptr && info != nullptr, "null argument");
  info[0] = p->siblings4;
  info[1] = p->row_siblings;
  info[2] = p->num_row_chunks;
  return TARL_OK;
}

extern "C" int tarl_plan_info(const tarl_plan* p, int64_t* info) {
  TARL_REQUIRE(p != nullptr && info != nullptr, "null argument");
  info[0] = p->N;
  info[1] = p->E;
  info[2] = p->G;
  info[3] = p->max_in;
  info[4] = p->max_out;
  info[5] = p->src_sorted;
  return TARL_OK;
}
