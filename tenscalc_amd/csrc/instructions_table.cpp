// Native scalar-instruction table: append / append-unique (CSE) / dependency export.
// From-scratch replacement for /root/reference/lib/csparse/instructionsTableUTHash.c
// (appendUniqueInstruction :534-619, getDependencies :746-768); see
// include/tc_instructions_table.h for the interface mapping.
#include "tc_instructions_table.h"

#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

struct Instr {
  int32_t type;
  int64_t opBegin, nOps;
  int64_t parBegin, nPars;
};

}  // namespace

struct tcit {
  std::vector<Instr> instrs;
  std::vector<int64_t> ops;
  std::vector<double> pars;
  // hash -> candidate instruction indices (0-based); collisions resolved by full compare
  std::unordered_multimap<uint64_t, int64_t> index;

  static uint64_t mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2);
    return h;
  }
  uint64_t hashOf(int32_t type, int64_t nP, const double *p, int64_t nO, const int64_t *o) const {
    uint64_t h = mix(0x2545F4914F6CDD1DULL, (uint64_t)type);
    h = mix(h, (uint64_t)nP);
    h = mix(h, (uint64_t)nO);
    for (int64_t i = 0; i < nO; i++) h = mix(h, (uint64_t)o[i]);
    for (int64_t i = 0; i < nP; i++) {
      uint64_t bits;
      std::memcpy(&bits, &p[i], sizeof bits);
      h = mix(h, bits);
    }
    return h;
  }
  bool same(const Instr &a, int32_t type, int64_t nP, const double *p, int64_t nO,
            const int64_t *o) const {
    if (a.type != type || a.nOps != nO || a.nPars != nP) return false;
    if (nO && std::memcmp(&ops[a.opBegin], o, sizeof(int64_t) * nO)) return false;
    if (nP && std::memcmp(&pars[a.parBegin], p, sizeof(double) * nP)) return false;
    return true;
  }
  int64_t push(int32_t type, int64_t nP, const double *p, int64_t nO, const int64_t *o) {
    Instr in{type, (int64_t)ops.size(), nO, (int64_t)pars.size(), nP};
    ops.insert(ops.end(), o, o + nO);
    pars.insert(pars.end(), p, p + nP);
    instrs.push_back(in);
    return (int64_t)instrs.size();  // 1-based
  }
};

extern "C" {

tcit_t *tcit_create(void) { return new tcit(); }
void tcit_destroy(tcit_t *t) { delete t; }

int64_t tcit_append(tcit_t *t, int32_t type, int64_t nP, const double *p, int64_t nO,
                    const int64_t *o) {
  if (!t || nP < 0 || nO < 0) return -1;
  for (int64_t i = 0; i < nO; i++)
    if (o[i] < 1 || o[i] > (int64_t)t->instrs.size()) return -1;  // operands must already exist
  int64_t idx = t->push(type, nP, p, nO, o);
  t->index.emplace(t->hashOf(type, nP, p, nO, o), idx - 1);
  return idx;
}

int64_t tcit_append_unique(tcit_t *t, int32_t type, int64_t nP, const double *p, int64_t nO,
                           const int64_t *o) {
  if (!t || nP < 0 || nO < 0) return -1;
  uint64_t h = t->hashOf(type, nP, p, nO, o);
  auto range = t->index.equal_range(h);
  for (auto it = range.first; it != range.second; ++it)
    if (t->same(t->instrs[it->second], type, nP, p, nO, o)) return it->second + 1;
  for (int64_t i = 0; i < nO; i++)
    if (o[i] < 1 || o[i] > (int64_t)t->instrs.size()) return -1;
  int64_t idx = t->push(type, nP, p, nO, o);
  t->index.emplace(h, idx - 1);
  return idx;
}

int tcit_append_many(tcit_t *t, int32_t type, int64_t n, int64_t nP, const double *p, int64_t nO,
                     const int64_t *o, int32_t unique, int64_t *out) {
  for (int64_t k = 0; k < n; k++) {
    int64_t r = unique ? tcit_append_unique(t, type, nP, p + k * nP, nO, o + k * nO)
                       : tcit_append(t, type, nP, p + k * nP, nO, o + k * nO);
    if (r < 0) return -1;
    out[k] = r;
  }
  return 0;
}

int tcit_get(tcit_t *t, int64_t index, int32_t *type, int64_t *nP, const double **p, int64_t *nO,
             const int64_t **o) {
  if (!t || index < 1 || index > (int64_t)t->instrs.size()) return -1;
  const Instr &in = t->instrs[index - 1];
  *type = in.type;
  *nP = in.nPars;
  *nO = in.nOps;
  *p = t->pars.data() + in.parBegin;
  *o = t->ops.data() + in.opBegin;
  return 0;
}

int64_t tcit_height(tcit_t *t) { return t ? (int64_t)t->instrs.size() : -1; }
int64_t tcit_total_operands(tcit_t *t) { return t ? (int64_t)t->ops.size() : -1; }
int64_t tcit_total_parameters(tcit_t *t) { return t ? (int64_t)t->pars.size() : -1; }

void tcit_dependencies(tcit_t *t, int64_t *children, int64_t *parents) {
  int64_t k = 0;
  for (size_t i = 0; i < t->instrs.size(); i++) {
    const Instr &in = t->instrs[i];
    for (int64_t j = 0; j < in.nOps; j++, k++) {
      children[k] = (int64_t)i + 1;
      parents[k] = t->ops[in.opBegin + j];
    }
  }
}

void tcit_find_by_type(tcit_t *t, int32_t type, int8_t *isOfType) {
  for (size_t i = 0; i < t->instrs.size(); i++) isOfType[i] = t->instrs[i].type == type;
}

void tcit_export(tcit_t *t, int32_t *types, int64_t *opStart, int64_t *ops, int64_t *parStart,
                 double *pars) {
  size_t n = t->instrs.size();
  for (size_t i = 0; i < n; i++) {
    types[i] = t->instrs[i].type;
    opStart[i] = t->instrs[i].opBegin;
    parStart[i] = t->instrs[i].parBegin;
  }
  opStart[n] = (int64_t)t->ops.size();
  parStart[n] = (int64_t)t->pars.size();
  if (!t->ops.empty()) std::memcpy(ops, t->ops.data(), sizeof(int64_t) * t->ops.size());
  if (!t->pars.empty()) std::memcpy(pars, t->pars.data(), sizeof(double) * t->pars.size());
}

}  // extern "C"
