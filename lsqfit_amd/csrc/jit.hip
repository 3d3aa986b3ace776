// The user's formula COMPILED instead of interpreted (gfx950, hiprtc).
//
// The reference differentiates an arbitrary Python fit function by pushing gvar.valder vectors
// through it (src/lsqfit/_gsl.pyx:742-760, `_c_df`); the device's stand-in is the RPN tape of
// lsqamd_set_tape.  Interpreting that tape (model.hip) costs ~40 scalar + vector + LDS instructions
// per tape instruction; here the tape is turned into straight-line HIP at lsqamd_set_tape time,
// built with hiprtc for gfx950 and launched in place of the interpreter (which stays as the route
// when hiprtc is not in the process / on the box, or when the formula is outside what the
// generator handles: lsqamd_debug_flags bit 3 says which one runs).
//
// Shape of the generated code.  The tape is parsed into its expression tree; every additive chain
// (+, -, unary minus) is flattened and its leaves are grouped by STRUCTURE (same operations,
// x's; only the parameter indices and literal constants differ).  A group of >= 16 look-alike terms -- the
// sum over states / exponentials / harmonics of a real fit function -- becomes ONE loop in which
// the 64 lanes of a wave stride over the terms of ONE data row: parameter loads and Jacobian
// stores are then contiguous 512-byte segments (families laid out p = [a_0.., w_0..]), exactly the
// access pattern of the hand-written sum kernels.  What is left (the "outer" expression: a few
// dozen nodes, its own parameters, the sums as inputs) is evaluated forward and differentiated in
// REVERSE by every lane; the adjoint of each sum then scales the term derivatives in a second loop
// -- or, when the sum is reached from the root through +/- only (adjoint +-1 known beforehand: the
// usual case), values and derivatives come out of a single loop.  Formulas without wide sums (the
// 27 NIST models) get one lane per data row and the whole expression in registers.
//
// Values in the generated code are computed with the same device functions as the hand-written
// kernels (devmath.h is embedded: sincos_moderate / cos_moderate) and libm otherwise.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
#include <iterator>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "fit_state.h"
#include "jit.h"

namespace {

const char *kDevmath =
#include "devmath_src.inc"
    ;

// ---- hiprtc, bound at run time ---------------------------------------------------------------------
typedef void *Prog;
struct Rtc {
  void *lib = nullptr;
  int (*Create)(Prog *, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*Compile)(Prog, int, const char **) = nullptr;
  int (*LogSize)(Prog, size_t *) = nullptr;
  int (*Log)(Prog, char *) = nullptr;
  int (*CodeSize)(Prog, size_t *) = nullptr;
  int (*Code)(Prog, char *) = nullptr;
  int (*Destroy)(Prog *) = nullptr;
  int (*Version)(int *, int *) = nullptr;
  std::string why, where;
  bool ok = false;
  int vmajor = 0, vminor = 0;
};

Rtc &rtc() {
  static Rtc r = [] {
    Rtc t;
    const char *off = getenv("LSQAMD_TAPE");
    if (off && (off[0] == 'i' || off[0] == 'w' || off[0] == 'f')) {   // interp / whole / forward: developer knobs
      t.why = "disabled by LSQAMD_TAPE";
      return t;
    }
    const char *path = getenv("LSQAMD_HIPRTC_PATH");
    std::vector<std::string> tries;
    if (path && *path) tries.push_back(path);
    // the copy that belongs to the HIP runtime THIS process runs on (inside PyTorch: torch/lib): a code
    // object built by another ROCm's compiler may not load
    Dl_info di;
    if (dladdr((void *)&hipModuleLoadData, &di) && di.dli_fname) {
      std::string dir = di.dli_fname;
      const size_t s = dir.rfind('/');
      if (s != std::string::npos) {
        dir.resize(s + 1);
        tries.push_back(dir + "libhiprtc.so");
        tries.push_back(dir + "libhiprtc.so.7");
      }
    }
    tries.push_back("libhiprtc.so");
    tries.push_back("libhiprtc.so.7");
    std::string errs;
    for (const std::string &n : tries) {
      t.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (t.lib) { t.where = n; break; }
      const char *e = dlerror();
      errs += (errs.empty() ? "" : "; ") + std::string(e ? e : "not found");
    }
    if (!t.lib) {
      t.why = "libhiprtc.so could not be loaded: " + errs;
      return t;
    }
    struct { const char *name; void **slot; } syms[] = {
        {"hiprtcCreateProgram", (void **)&t.Create}, {"hiprtcCompileProgram", (void **)&t.Compile},
        {"hiprtcGetProgramLogSize", (void **)&t.LogSize}, {"hiprtcGetProgramLog", (void **)&t.Log},
        {"hiprtcGetCodeSize", (void **)&t.CodeSize}, {"hiprtcGetCode", (void **)&t.Code},
        {"hiprtcDestroyProgram", (void **)&t.Destroy}, {"hiprtcVersion", (void **)&t.Version}};
    for (auto &s : syms) {
      *s.slot = dlsym(t.lib, s.name);
      if (!*s.slot) {
        t.why = std::string("libhiprtc.so lacks ") + s.name;
        return t;
      }
    }
    (void)t.Version(&t.vmajor, &t.vminor);
    t.ok = true;
    return t;
  }();
  return r;
}

// ---- the tape as a tree ----------------------------------------------------------------------------
enum { OP_WSUM = 100 };   // pseudo-node of the outer expression: the value of a wide sum
struct Node {
  int op = 0, arg = 0, a = -1, b = -1;
};

struct Group {            // look-alike terms of one additive chain, one sign
  int sign = 1;
  int tmpl = -1;          // root (in the ORIGINAL node array) of the first term: the template the loop body is made from
  int n_terms = 0, n_slots = 0;
  std::vector<std::vector<int>> idx;   // [slot][term] parameter index
  std::vector<int> shared;             // [slot] 1: the same parameter in every term (accumulated), 0: private (stored)
  std::vector<int> base, stride;       // [slot] affine index base + k * stride, or stride = INT_MIN: table
  std::vector<int> cnodes;             // constant nodes of the template term, in traversal order
  std::vector<std::vector<double>> cvals;   // [constant][term]; one entry when the value is the same in every term
};
struct WSum {
  std::vector<Group> groups;
  int fused = 0;          // adjoint known beforehand: +1 / -1 (one loop), 0: from the outer reverse sweep (two loops)
};
struct Plan {
  std::vector<Node> nodes;      // original tree
  int root = -1;
  std::vector<Node> outer;      // outer expression (WSUM pseudo-nodes; P / X / CONST leaves as in the original)
  int oroot = -1;
  std::vector<WSum> wsums;
  std::vector<int> out_params;  // parameters whose derivative is formed by the outer sweep (incl. shared slots)
  std::vector<int> unread;      // parameters the formula never reads: their Jacobian column is zero
  std::vector<double> consts;
  int P = 0, n_x = 1;
  bool wave_per_row = false;
  bool nrm_ok = false;          // few parameters, no wide sums: a third kernel forms J^T J, J^T f and chi2 without writing J
  bool fit_ok = false;          // ... and the whole-fit kernels (lsqamd_jit_lm / _lmb): up to FIT_MAX_P parameters
};

bool is_push(int op) { return op <= LSQAMD_OP_P; }
bool is_bin(int op) { return op >= LSQAMD_OP_ADD && op <= LSQAMD_OP_POW; }

bool parse_tape(const int32_t *code, int n, std::vector<Node> &nodes, int &root) {
  std::vector<int> st;
  nodes.clear();
  nodes.reserve((size_t)n);
  for (int t = 0; t < n; ++t) {
    Node nd;
    nd.op = code[t] & 0xff;
    nd.arg = code[t] >> 8;
    if (is_push(nd.op)) {
    } else if (is_bin(nd.op)) {
      if (st.size() < 2) return false;
      nd.b = st.back(); st.pop_back();
      nd.a = st.back(); st.pop_back();
    } else {
      if (st.empty()) return false;
      nd.a = st.back(); st.pop_back();
    }
    nodes.push_back(nd);
    st.push_back((int)nodes.size() - 1);
  }
  if (st.size() != 1) return false;
  root = st[0];
  return true;
}

bool additive(int op) { return op == LSQAMD_OP_ADD || op == LSQAMD_OP_SUB; }

// leaves (node, sign) of the additive chain rooted at n, left to right
void flatten(const std::vector<Node> &nodes, int n, std::vector<std::pair<int, int>> &leaves) {
  std::vector<std::pair<int, int>> st{{n, 1}};
  while (!st.empty()) {
    auto [k, s] = st.back();
    st.pop_back();
    const Node &nd = nodes[(size_t)k];
    if (nd.op == LSQAMD_OP_ADD) { st.push_back({nd.b, s}); st.push_back({nd.a, s}); }
    else if (nd.op == LSQAMD_OP_SUB) { st.push_back({nd.b, -s}); st.push_back({nd.a, s}); }
    else if (nd.op == LSQAMD_OP_NEG && (additive(nodes[(size_t)nd.a].op) || nodes[(size_t)nd.a].op == LSQAMD_OP_NEG)) st.push_back({nd.a, -s});
    else leaves.push_back({k, s});
  }
}

// structure of a subtree with the parameter indices abstracted into slots (first-use order)
struct Sig {
  std::string text;
  std::vector<int> params;   // slot -> parameter index
  std::vector<double> cvals; // constants in traversal order (their VALUES are not part of the structure: a Fourier series
                             // a_k cos(k x) with literal k is one group with a table of k's)
  std::vector<int> cnodes;   // ... and the nodes they sit in
  int n_nodes = 0;
};
void signature(const std::vector<Node> &nodes, const std::vector<double> &consts, int n, Sig &s) {
  const Node &nd = nodes[(size_t)n];
  ++s.n_nodes;
  char b[64];
  if (nd.op == LSQAMD_OP_P) {
    int slot = -1;
    for (size_t i = 0; i < s.params.size(); ++i)
      if (s.params[i] == nd.arg) slot = (int)i;
    if (slot < 0) { slot = (int)s.params.size(); s.params.push_back(nd.arg); }
    snprintf(b, sizeof(b), "P%d;", slot);
  } else if (nd.op == LSQAMD_OP_CONST) {
    snprintf(b, sizeof(b), "K%d;", (int)s.cvals.size());
    s.cvals.push_back(consts[(size_t)nd.arg]);
    s.cnodes.push_back(n);
  } else if (nd.op == LSQAMD_OP_X) {
    snprintf(b, sizeof(b), "X%d;", nd.arg);
  } else {
    if (nd.a >= 0) signature(nodes, consts, nd.a, s);
    if (nd.b >= 0) signature(nodes, consts, nd.b, s);
    snprintf(b, sizeof(b), "o%d,%d;", nd.op, nd.op == LSQAMD_OP_POWI ? nd.arg : 0);
  }
  s.text += b;
}

int subtree_nodes(const std::vector<Node> &nodes, int n) {
  int c = 0;
  std::vector<int> st{n};
  while (!st.empty()) {
    const int k = st.back();
    st.pop_back();
    ++c;
    if (nodes[(size_t)k].a >= 0) st.push_back(nodes[(size_t)k].a);
    if (nodes[(size_t)k].b >= 0) st.push_back(nodes[(size_t)k].b);
  }
  return c;
}

// (a group keeps n_terms of a wave's 64 lanes busy: below a quarter of them the terms are better off unrolled in the
// one-lane-per-row form -- a sum of 2-6 exponentials, lsqfit's canonical fit, then also gets the register-resident
// normal equations)
constexpr int MIN_GROUP = 16, MAX_TERM_NODES = 160, MAX_TERM_SLOTS = 16, MAX_OUTER_NODES = 768, MAX_OUT_PARAMS = 64;

struct Builder {
  Plan &pl;
  std::vector<int> reads;      // reads of every parameter on the whole tape
  bool too_deep = false;
  explicit Builder(Plan &p) : pl(p) {}

  int add_outer(const Node &nd) {
    pl.outer.push_back(nd);
    return (int)pl.outer.size() - 1;
  }

  // original node n -> outer node (wide sums become pseudo-nodes)
  int xf(int n, int depth) {
    if (depth > 200) { too_deep = true; return add_outer(Node()); }
    const Node &nd = pl.nodes[(size_t)n];
    if (!additive(nd.op)) {
      Node o = nd;
      if (nd.a >= 0) o.a = xf(nd.a, depth + 1);
      if (nd.b >= 0) o.b = xf(nd.b, depth + 1);
      return add_outer(o);
    }
    std::vector<std::pair<int, int>> leaves;
    flatten(pl.nodes, n, leaves);
    // group look-alike leaves
    std::map<std::string, std::vector<size_t>> by_sig;
    std::vector<Sig> sigs(leaves.size());
    for (size_t i = 0; i < leaves.size(); ++i) {
      if (subtree_nodes(pl.nodes, leaves[i].first) > MAX_TERM_NODES) continue;
      signature(pl.nodes, pl.consts, leaves[i].first, sigs[i]);
      if (sigs[i].params.empty() || (int)sigs[i].params.size() > MAX_TERM_SLOTS) continue;
      by_sig[(leaves[i].second > 0 ? "+" : "-") + sigs[i].text].push_back(i);
    }
    std::vector<char> grouped(leaves.size(), 0);
    WSum ws;
    for (auto &kv : by_sig) {
      const std::vector<size_t> &mem = kv.second;
      if ((int)mem.size() < MIN_GROUP) continue;
      Group g;
      g.sign = leaves[mem[0]].second;
      g.tmpl = leaves[mem[0]].first;
      g.n_terms = (int)mem.size();
      g.n_slots = (int)sigs[mem[0]].params.size();
      g.idx.assign((size_t)g.n_slots, std::vector<int>((size_t)g.n_terms));
      for (int k = 0; k < g.n_terms; ++k)
        for (int s = 0; s < g.n_slots; ++s) g.idx[(size_t)s][(size_t)k] = sigs[mem[(size_t)k]].params[(size_t)s];
      // reads of each parameter inside ONE term (same for every term: same structure)
      std::vector<int> in_term((size_t)g.n_slots, 0);
      {
        std::vector<int> st{g.tmpl};
        while (!st.empty()) {
          const Node &t = pl.nodes[(size_t)st.back()];
          st.pop_back();
          if (t.op == LSQAMD_OP_P)
            for (int s = 0; s < g.n_slots; ++s)
              if (g.idx[(size_t)s][0] == t.arg) ++in_term[(size_t)s];
          if (t.a >= 0) st.push_back(t.a);
          if (t.b >= 0) st.push_back(t.b);
        }
      }
      bool ok = true;
      g.shared.assign((size_t)g.n_slots, 0);
      g.base.assign((size_t)g.n_slots, 0);
      g.stride.assign((size_t)g.n_slots, 0);
      for (int s = 0; s < g.n_slots && ok; ++s) {
        const std::vector<int> &ix = g.idx[(size_t)s];
        bool all_same = true, priv = true;
        for (int k = 1; k < g.n_terms; ++k) all_same = all_same && ix[(size_t)k] == ix[0];
        if (all_same) { g.shared[(size_t)s] = 1; g.base[(size_t)s] = ix[0]; g.stride[(size_t)s] = 0; continue; }
        for (int k = 0; k < g.n_terms && priv; ++k) priv = reads[(size_t)ix[(size_t)k]] == in_term[(size_t)s];
        if (!priv) { ok = false; break; }   // read elsewhere as well (or by several terms): not a plain store
        bool affine = true;
        const int st0 = ix[1] - ix[0];
        for (int k = 2; k < g.n_terms; ++k) affine = affine && ix[(size_t)k] - ix[(size_t)k - 1] == st0;
        g.base[(size_t)s] = ix[0];
        g.stride[(size_t)s] = affine ? st0 : INT32_MIN;
      }
      if (!ok) continue;
      g.cnodes = sigs[mem[0]].cnodes;
      g.cvals.assign(g.cnodes.size(), std::vector<double>());
      for (size_t c = 0; c < g.cnodes.size(); ++c) {
        bool same = true;
        for (int k = 1; k < g.n_terms; ++k) {
          uint64_t u0, u1;
          std::memcpy(&u0, &sigs[mem[0]].cvals[c], 8);
          std::memcpy(&u1, &sigs[mem[(size_t)k]].cvals[c], 8);
          same = same && u0 == u1;
        }
        if (same) g.cvals[c].push_back(sigs[mem[0]].cvals[c]);
        else
          for (int k = 0; k < g.n_terms; ++k) g.cvals[c].push_back(sigs[mem[(size_t)k]].cvals[c]);
      }
      for (size_t m : mem) grouped[m] = 1;
      ws.groups.push_back(std::move(g));
    }
    int cur = -1;
    if (!ws.groups.empty()) {
      Node o;
      o.op = OP_WSUM;
      o.arg = (int)pl.wsums.size();
      pl.wsums.push_back(std::move(ws));
      cur = add_outer(o);
    }
    for (size_t i = 0; i < leaves.size(); ++i) {
      if (grouped[i]) continue;
      const int leaf = xf(leaves[i].first, depth + 1);
      if (cur < 0) {
        if (leaves[i].second > 0) cur = leaf;
        else { Node o; o.op = LSQAMD_OP_NEG; o.a = leaf; cur = add_outer(o); }
      } else {
        Node o;
        o.op = leaves[i].second > 0 ? LSQAMD_OP_ADD : LSQAMD_OP_SUB;
        o.a = cur;
        o.b = leaf;
        cur = add_outer(o);
      }
    }
    return cur;
  }
};

bool make_plan(const int32_t *code, int n_code, const double *consts, int n_consts, int P, int n_x, Plan &pl, std::string &why) {
  pl.P = P;
  pl.n_x = n_x < 1 ? 1 : n_x;
  pl.consts.assign(consts, consts + (n_consts > 0 ? n_consts : 0));
  if (!parse_tape(code, n_code, pl.nodes, pl.root)) { why = "malformed tape"; return false; }
  Builder b(pl);
  b.reads.assign((size_t)P, 0);
  for (const Node &nd : pl.nodes)
    if (nd.op == LSQAMD_OP_P) ++b.reads[(size_t)nd.arg];
  for (int j = 0; j < P; ++j)
    if (b.reads[(size_t)j] == 0) pl.unread.push_back(j);
  pl.oroot = b.xf(pl.root, 0);
  if (b.too_deep) { why = "expression nested too deeply"; return false; }
  if ((int)pl.outer.size() > MAX_OUTER_NODES) { why = "the part of the formula outside its wide sums is too large"; return false; }
  // adjoints known beforehand: sums reached from the root through + / - / unary minus only
  {
    std::vector<std::pair<int, int>> st{{pl.oroot, 1}};
    while (!st.empty()) {
      auto [k, s] = st.back();
      st.pop_back();
      const Node &nd = pl.outer[(size_t)k];
      if (nd.op == OP_WSUM) pl.wsums[(size_t)nd.arg].fused = s;
      else if (nd.op == LSQAMD_OP_ADD) { st.push_back({nd.a, s}); st.push_back({nd.b, s}); }
      else if (nd.op == LSQAMD_OP_SUB) { st.push_back({nd.a, s}); st.push_back({nd.b, -s}); }
      else if (nd.op == LSQAMD_OP_NEG) st.push_back({nd.a, -s});
    }
  }
  // parameters differentiated by the outer sweep: its own leaves + the shared slots of the groups
  std::vector<char> isout((size_t)P, 0);
  for (const Node &nd : pl.outer)
    if (nd.op == LSQAMD_OP_P) isout[(size_t)nd.arg] = 1;
  for (const WSum &w : pl.wsums)
    for (const Group &g : w.groups)
      for (int s = 0; s < g.n_slots; ++s)
        if (g.shared[(size_t)s]) isout[(size_t)g.base[(size_t)s]] = 1;
  for (int j = 0; j < P; ++j)
    if (isout[(size_t)j]) pl.out_params.push_back(j);
  if ((int)pl.out_params.size() > MAX_OUT_PARAMS) { why = "too many parameters outside the wide sums"; return false; }
  pl.wave_per_row = !pl.wsums.empty();
  pl.nrm_ok = !pl.wave_per_row && P >= 1 && P <= lsqamd_jit::NRM_MAX_P;
  pl.fit_ok = !pl.wave_per_row && P >= 1 && P <= lsqamd_jit::FIT_MAX_P;
  return true;
}

// ---- code generation -------------------------------------------------------------------------------
struct Src {
  std::string s;
  void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
    char b[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(b, sizeof(b), fmt, ap);
    va_end(ap);
    s += b;
  }
};

std::string dlit(double v) {   // exact literal
  char b[64];
  if (std::isnan(v)) return "__builtin_nan(\"\")";
  if (std::isinf(v)) return v > 0 ? "__builtin_inf()" : "(-__builtin_inf())";
  snprintf(b, sizeof(b), "%a", v);
  return b;
}

// Emits forward values (and, jac, local partials) of a tree, then its reverse sweep.  Variables are named
// <pfx>v<node> / <pfx>d<node>a|b / <pfx>g<node>.
struct TreeGen {
  const std::vector<Node> &nodes;
  const Plan &pl;
  Src &o;
  std::string pfx, ind;
  bool jac;
  std::function<std::string(const Node &)> leaf_value;                          // P / WSUM leaves
  std::function<void(const Node &, const std::string &)> leaf_adjoint;          // adjoint arriving at a P / WSUM leaf
  std::function<std::string(int)> const_value;                                  // optional: expression for the constant in node n
  TreeGen(const std::vector<Node> &n, const Plan &p, Src &out, const std::string &prefix, const std::string &indent, bool j)
      : nodes(n), pl(p), o(out), pfx(prefix), ind(indent), jac(j) {}

  std::string v(int n) const { return pfx + "v" + std::to_string(n); }
  std::string d(int n, char w) const { return pfx + "d" + std::to_string(n) + w; }

  void forward(int n) {
    const Node &nd = nodes[(size_t)n];
    if (nd.a >= 0) forward(nd.a);
    if (nd.b >= 0) forward(nd.b);
    const std::string V = v(n), I = ind;
    const char *i = I.c_str();
    const std::string A = nd.a >= 0 ? v(nd.a) : "", B = nd.b >= 0 ? v(nd.b) : "";
    const char *a = A.c_str(), *b = B.c_str(), *vv = V.c_str();
    switch (nd.op) {
      case LSQAMD_OP_CONST:
        o.f("%sconst double %s = %s;\n", i, vv, (const_value ? const_value(n) : dlit(pl.consts[(size_t)nd.arg])).c_str());
        break;
      case LSQAMD_OP_X: o.f("%sconst double %s = x%d;\n", i, vv, nd.arg); break;
      case LSQAMD_OP_P:
      case OP_WSUM: o.f("%sconst double %s = %s;\n", i, vv, leaf_value(nd).c_str()); break;
      case LSQAMD_OP_ADD: o.f("%sconst double %s = %s + %s;\n", i, vv, a, b); break;
      case LSQAMD_OP_SUB: o.f("%sconst double %s = %s - %s;\n", i, vv, a, b); break;
      case LSQAMD_OP_MUL: o.f("%sconst double %s = %s * %s;\n", i, vv, a, b); break;
      case LSQAMD_OP_DIV:
        o.f("%sconst double %s = %s / %s;\n", i, vv, a, b);
        if (jac) o.f("%sconst double %s = 1.0 / %s, %s = -%s / %s;\n", i, d(n, 'a').c_str(), b, d(n, 'b').c_str(), vv, b);
        break;
      case LSQAMD_OP_POW:
        o.f("%sconst double %s = pow(%s, %s);\n", i, vv, a, b);
        if (jac)
          o.f("%sconst double %s = %s * pow(%s, %s - 1.0), %s = (%s > 0.0) ? %s * log(%s) : 0.0;\n", i, d(n, 'a').c_str(), b, a, b,
              d(n, 'b').c_str(), a, vv, a);
        break;
      case LSQAMD_OP_NEG: o.f("%sconst double %s = -%s;\n", i, vv, a); break;
      case LSQAMD_OP_EXP: o.f("%sconst double %s = exp(%s);\n", i, vv, a); break;
      case LSQAMD_OP_LOG:
        o.f("%sconst double %s = log(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %s = 1.0 / %s;\n", i, d(n, 'a').c_str(), a);
        break;
      case LSQAMD_OP_SIN:
        if (jac) o.f("%sdouble %s, %s; sincos_moderate<true>(%s, &%s, &%s);\n", i, vv, d(n, 'a').c_str(), a, vv, d(n, 'a').c_str());
        else o.f("%sdouble %s, %sc; sincos_moderate<true>(%s, &%s, &%sc);\n", i, vv, vv, a, vv, vv);
        break;
      case LSQAMD_OP_COS:
        if (jac) o.f("%sdouble %s, %ss; sincos_moderate<true>(%s, &%ss, &%s); const double %s = -%ss;\n", i, vv, vv, a, vv, vv, d(n, 'a').c_str(), vv);
        else o.f("%sconst double %s = cos_moderate<true>(%s);\n", i, vv, a);
        break;
      case LSQAMD_OP_ATAN:
        o.f("%sconst double %s = atan(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %s = 1.0 / (1.0 + %s * %s);\n", i, d(n, 'a').c_str(), a, a);
        break;
      case LSQAMD_OP_SQRT:
        o.f("%sconst double %s = sqrt(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %s = 0.5 / %s;\n", i, d(n, 'a').c_str(), vv);
        break;
      case LSQAMD_OP_POWI:
        if (nd.arg == 0) { o.f("%sconst double %s = 1.0;\n", i, vv); if (jac) o.f("%sconst double %s = 0.0;\n", i, d(n, 'a').c_str()); }
        else if (nd.arg == 1) { o.f("%sconst double %s = %s;\n", i, vv, a); if (jac) o.f("%sconst double %s = 1.0;\n", i, d(n, 'a').c_str()); }
        else if (nd.arg == 2) { o.f("%sconst double %s = %s * %s;\n", i, vv, a, a); if (jac) o.f("%sconst double %s = 2.0 * %s;\n", i, d(n, 'a').c_str(), a); }
        else {
          o.f("%sconst double %s = pow(%s, %d.0);\n", i, vv, a, nd.arg);
          if (jac) o.f("%sconst double %s = %d.0 * pow(%s, %d.0);\n", i, d(n, 'a').c_str(), nd.arg, a, nd.arg - 1);
        }
        break;
      // (the formulas of common.h tape_unary_ext)
      case LSQAMD_OP_TAN:
        o.f("%sconst double %s = tan(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %s = 1.0 + %s * %s;\n", i, d(n, 'a').c_str(), vv, vv);
        break;
      case LSQAMD_OP_SINH:
        o.f("%sconst double %s = sinh(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %s = cosh(%s);\n", i, d(n, 'a').c_str(), a);
        break;
      case LSQAMD_OP_COSH:
        o.f("%sconst double %s = cosh(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %s = sinh(%s);\n", i, d(n, 'a').c_str(), a);
        break;
      case LSQAMD_OP_TANH:
        o.f("%sconst double %s = tanh(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %se = exp(-2.0 * fabs(%s)), %s = 4.0 * %se / ((1.0 + %se) * (1.0 + %se));\n", i, vv, a, d(n, 'a').c_str(), vv, vv, vv);
        break;
      case LSQAMD_OP_ASIN:
        o.f("%sconst double %s = asin(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %s = 1.0 / sqrt(1.0 - %s * %s);\n", i, d(n, 'a').c_str(), a, a);
        break;
      case LSQAMD_OP_ACOS:
        o.f("%sconst double %s = acos(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %s = -1.0 / sqrt(1.0 - %s * %s);\n", i, d(n, 'a').c_str(), a, a);
        break;
      case LSQAMD_OP_ABS:
        o.f("%sconst double %s = fabs(%s);\n", i, vv, a);
        if (jac) o.f("%sconst double %s = %s >= 0.0 ? 1.0 : -1.0;\n", i, d(n, 'a').c_str(), a);
        break;
      default: o.f("%sconst double %s = %s;\n", i, vv, a); if (jac) o.f("%sconst double %s = 1.0;\n", i, d(n, 'a').c_str()); break;
    }
  }

  // adjoint `g` (an identifier or literal) arrives at node n
  void reverse(int n, const std::string &g) {
    const Node &nd = nodes[(size_t)n];
    const char *i = ind.c_str();
    auto child = [&](int c, const std::string &expr) {   // name the child's adjoint, then descend
      const Node &cn = nodes[(size_t)c];
      if (cn.op == LSQAMD_OP_CONST || cn.op == LSQAMD_OP_X) return;
      const std::string G = pfx + "g" + std::to_string(c);
      o.f("%sconst double %s = %s;\n", i, G.c_str(), expr.c_str());
      reverse(c, G);
    };
    switch (nd.op) {
      case LSQAMD_OP_CONST:
      case LSQAMD_OP_X: break;
      case LSQAMD_OP_P:
      case OP_WSUM: leaf_adjoint(nd, g); break;
      case LSQAMD_OP_ADD: child(nd.a, g); child(nd.b, g); break;
      case LSQAMD_OP_SUB: child(nd.a, g); child(nd.b, "-" + g); break;
      case LSQAMD_OP_MUL: child(nd.a, g + " * " + v(nd.b)); child(nd.b, g + " * " + v(nd.a)); break;
      case LSQAMD_OP_DIV:
      case LSQAMD_OP_POW: child(nd.a, g + " * " + d(n, 'a')); child(nd.b, g + " * " + d(n, 'b')); break;
      case LSQAMD_OP_NEG: child(nd.a, "-" + g); break;
      case LSQAMD_OP_EXP: child(nd.a, g + " * " + v(n)); break;
      default: child(nd.a, g + " * " + d(n, 'a')); break;
    }
  }
};

std::string slot_index(const Group &g, int gi_global, int s, const char *k) {
  char b[96];
  if (g.shared[(size_t)s]) snprintf(b, sizeof(b), "%d", g.base[(size_t)s]);
  else if (g.stride[(size_t)s] == INT32_MIN) snprintf(b, sizeof(b), "T%d_%d[%s]", gi_global, s, k);
  else if (g.stride[(size_t)s] == 1) snprintf(b, sizeof(b), "(%d + %s)", g.base[(size_t)s], k);
  else snprintf(b, sizeof(b), "(%d + %s * %d)", g.base[(size_t)s], k, g.stride[(size_t)s]);
  return b;
}

// one loop over the terms of a group.  mode 0: values only; 1: values + derivatives (adjoint `adj`, a literal or a
// variable); 2: derivatives only (second loop of a sum whose adjoint came out of the outer sweep)
void gen_group(const Plan &pl, Src &o, const Group &g, int gid, int mode, const std::string &adj, const std::string &acc) {
  const bool der = mode >= 1;
  o.f("      for (int k = lane; k < %d; k += 64) {\n", g.n_terms);
  for (int s = 0; s < g.n_slots; ++s) o.f("        const double q%d = sp[%s];\n", s, slot_index(g, gid, s, "k").c_str());
  TreeGen tg(pl.nodes, pl, o, "t", "        ", der);
  tg.leaf_value = [&](const Node &nd) {
    for (int s = 0; s < g.n_slots; ++s)
      if (g.idx[(size_t)s][0] == nd.arg) return "q" + std::to_string(s);
    return std::string("0.0");
  };
  tg.const_value = [&](int node) {
    for (size_t c = 0; c < g.cnodes.size(); ++c)
      if (g.cnodes[c] == node) {
        if (g.cvals[c].size() == 1) return dlit(g.cvals[c][0]);
        return "CT" + std::to_string(gid) + "_" + std::to_string(c) + "[k]";
      }
    return dlit(pl.consts[(size_t)pl.nodes[(size_t)node].arg]);
  };
  if (der)
    for (int s = 0; s < g.n_slots; ++s) o.f("        double e%d = 0.0;\n", s);
  tg.leaf_adjoint = [&](const Node &nd, const std::string &gg) {
    for (int s = 0; s < g.n_slots; ++s)
      if (g.idx[(size_t)s][0] == nd.arg) o.f("        e%d += %s;\n", s, gg.c_str());
  };
  tg.forward(g.tmpl);
  if (mode <= 1) o.f("        %s += %s;\n", acc.c_str(), tg.v(g.tmpl).c_str());
  if (der) {
    o.f("        const double tadj = %s%s;\n", g.sign > 0 ? "" : "-", adj.c_str());
    tg.reverse(g.tmpl, "tadj");
    for (int s = 0; s < g.n_slots; ++s) {
      if (g.shared[(size_t)s]) {
        int j = 0;
        while (pl.out_params[(size_t)j] != g.base[(size_t)s]) ++j;
        o.f("        sh%d += e%d;\n", j, s);
      } else {
        o.f("        dst[%s] = w * e%d;\n", slot_index(g, gid, s, "k").c_str(), s);
      }
    }
  }
  o.f("      }\n");
}


// ---- a whole small fit in ONE launch (lsqamd_jit_lm): the text below follows the LM_* functions generate() emits -------
// Plain Levenberg-Marquardt (what the reference gets from gsl_multifit_nlinear_init / _driver / _covar, src/lsqfit/_gsl.pyx:
// 676-677,:706, with trs = lm) as api.hip iterate_device runs it over half a dozen launches per iteration -- the same solve
// (vecops.hip lm_tiny12_solve_kernel), the same decision (lm_trial_tail_small_kernel), the same scaling update and
// convergence test (lm_accept_tail_kernel) -- by one workgroup that keeps x, D, J^T J, J^T f in LDS from the first
// evaluation to the last.  Anything irregular (no positive pivot, a step that is not finite, a pivot that retained too
// little of its column under solver = qr, no progress in the very first iteration, a chi2 that is not finite) ends the kernel with
// reason 2 and the host runs the fit through the general path from the start.
const char *kLmDriver = R"LSQLM(
struct LmArgs {
  const double *x, *ymean, *wdiag; long long n_data;
  const unsigned char *in_block; const double *wt; const long long *blk_row0, *blk_size, *blk_woff; long long n_blocks;
  const double *p0;
  double *p, *p_trial, *dscale, *apk, *gvec, *v_out, *coln2, *st;
  const double *prior_prec, *prior_mean;
  int prior_dense, scaler, maxit, watch;
  double xtol, gtol, factor_up, factor_down, hostptr_bits;
  double *cov; long long ldc; int want_cov, pad_;     // (A)^-1 at the end point: cov[i * ldc + j], when want_cov
  double *host;      // record block (DEVICE memory): [0, 16) the record, [16, 24) reason nit nfev njev ntrial cov-formed logdet -, then x g D coln2 v (LP + 1 each)
  unsigned long long *pub, seq;   // pub != 0: device-visible HOST block the finished record block is published to (see the end of lm_fit)
};

static __device__ void m_diag(double *o, long long a0, long long a1, long long a2, long long a3, long long a4) {
  o[0] = (double)a0; o[1] = (double)a1; o[2] = (double)a2; o[3] = (double)a3; o[4] = (double)a4;
}

static __device__ void lm_normal(const LmArgs &a, const double *sp, double *red, double *sq, double *sA, double *sG, double *sT, double *ss,
                                 double *srow, const int *sblk) {
  const int tid = threadIdx.x;
  lm_nrm(a, sp, red, sq, srow, sblk);
  for (int e = tid; e < LP * LP; e += 256) {
    const int i = e / LP, j = e % LP;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    double v = sq[lo * LP - lo * (lo - 1) / 2 + (hi - lo)];
    if (a.prior_prec) v += a.prior_dense ? a.prior_prec[i * LP + j] : (i == j ? a.prior_prec[i] : 0.0);
    sA[e] = v;
  }
  if (tid < LP) {
    double t = 0.0;
    if (a.prior_prec) {
      if (a.prior_dense) { for (int k = 0; k < LP; ++k) t += a.prior_prec[tid * LP + k] * (sp[k] - a.prior_mean[k]); }
      else t = a.prior_prec[tid] * (sp[tid] - a.prior_mean[tid]);
    }
    sT[tid] = t;
    sG[tid] = sq[LNA + tid] + t;
  }
  __syncthreads();
  if (tid == 0) {
    double c = sq[LNA + LP];
    if (a.prior_prec) for (int k = 0; k < LP; ++k) c += (sp[k] - a.prior_mean[k]) * sT[k];
    ss[S_CHI2] = c;
  }
  __syncthreads();
}

// (A + mu D^2) v = g by wave 0: lane j holds column j of the upper triangle (rows i <= j), lane LP the right-hand side.
// Every cross-lane read names its lane at compile time (v_readlane: the value arrives in scalar registers), pivots are
// inverted by v_rsq_f64 + two Newton steps instead of a square root and a division, and the back substitution runs on a
// wave-uniform copy of y -- the dependent chain of a step is a dozen instructions.
static __device__ __forceinline__ double lm_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = -0.5 * d;
  double e = __builtin_fma(h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(h * y, y, 0.5);
  return __builtin_fma(y, e, y);
}

static __device__ void lm_solve(const double *sA, const double *sG, const double *sD, const double *sp, double mu, int watch,
                                double *sV, double *spt, double *ss, int *si) {
  const int lane = threadIdx.x;
  double m[LP], uinv[LP];
  double diag0 = 1.0;
  // (branch-free: every lane reads a clamped address, the selection follows)
  const int col = lane < LP ? lane : LP - 1;
  const double dl = sD[col], dmu = mu * dl * dl;
#pragma unroll
  for (int i = 0; i < LP; ++i) {
    const double av = sA[i * LP + col], gv = sG[i];
    double v = (lane < LP && i <= lane) ? av : 0.0;
    if (i == lane) { v += dmu; diag0 = v; }
    m[i] = lane == LP ? gv : v;
  }
  int fail = 0;
  double pmin = __builtin_huge_val();
#pragma unroll
  for (int k = 0; k < LP; ++k) {
    const double pk = lm_rl(m[k], k);
    if (!(pk > 0.0) && fail == 0) fail = k + 1;
    const double inv = lm_rsqrt(pk > 0.0 ? pk : 1.0);
    uinv[k] = inv;
    if (lane == k) pmin = pk;               // (this lane's pivot; divided by the damped diagonal entry below)
    m[k] *= inv;                            // (unconditional, like the updates: what lands below the diagonal is never read)
#pragma unroll
    for (int i = k + 1; i < LP; ++i) m[i] = __builtin_fma(-lm_rl(m[k], i), m[k], m[i]);
  }
  // U v = y: y (lane LP's column) wave-uniform, v_k = y_k / U_kk, then y_i -= U_ik v_k for the rows above
  // (the compiler sees that the back substitution broadcasts what the sweep broadcast already and keeps all of it alive in
  //  spilled scalar registers -- two v_writelane and two v_readlane per value: cheaper to broadcast again)
#pragma unroll
  for (int i = 0; i < LP; ++i) asm volatile("" : "+v"(m[i]));
  double y[LP], v[LP];
#pragma unroll
  for (int i = 0; i < LP; ++i) y[i] = lm_rl(m[i], LP);
#pragma unroll
  for (int k = LP - 1; k >= 0; --k) {
    v[k] = y[k] * uinv[k];
#pragma unroll
    for (int i = 0; i < k; ++i) y[i] -= lm_rl(m[i], k) * v[k];
  }
  const bool bad = fail != 0;
  // v is wave-uniform: its dot products as a handful of uniform FMAs (no cross-lane reduction), lane k keeps v_k
  double vl = 0.0, vg = 0.0, dv2 = 0.0, nf = 0.0;
#pragma unroll
  for (int k = 0; k < LP; ++k) {
    vl = lane == k ? v[k] : vl;
    vg = __builtin_fma(v[k], sG[k], vg);
    const double t = sD[k] * v[k];
    dv2 = __builtin_fma(t, t, dv2);
    nf += (v[k] - v[k] == 0.0) ? 0.0 : 1.0;
  }
  if (bad) { vl = __builtin_nan(""); nf = 1.0; }
  if (lane < LP) {
    sV[lane] = vl;
    spt[lane] = sp[lane] - vl;
    pmin = pmin / diag0;                    // the share of column `lane` its pivot retained
  } else {
    pmin = __builtin_huge_val();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pmin = fmin(pmin, __shfl_xor(pmin, o, 64));
  if (lane == 0) {
    si[0] = fail;
    ss[S_VG] = vg;
    ss[S_DV2] = dv2;
    ss[S_VFINITE] = nf == 0.0 ? 1.0 : 0.0;
    ss[S_PIVMIN] = (watch && !bad) ? pmin : 1.0;
  }
}

// (J^T J + prior)^-1 and its log det at the end point by wave 0: the elimination of lm_solve with the LP columns of the
// identity as right-hand sides in lanes LP .. 2 LP - 1 (they are forward-substituted along the way), then one back
// substitution per lane.  -> 0, or the failed pivot + 1 (the host then takes the general route, which knows what to do
// with a rank-deficient matrix)
static __device__ int lm_cov(const double *sA, double *cov, long long ldc, double *hcov, double *logdet) {
  const int lane = threadIdx.x;
  double m[LP], uinv[LP];
  const int col = lane < LP ? lane : LP - 1;
#pragma unroll
  for (int i = 0; i < LP; ++i) {
    const double av = sA[i * LP + col];
    double v = (lane < LP && i <= lane) ? av : 0.0;
    if (lane == LP + i) v = 1.0;
    m[i] = v;
  }
  int fail = 0;
  double ld = 0.0;
#pragma unroll
  for (int k = 0; k < LP; ++k) {
    const double pk = lm_rl(m[k], k);
    if (!(pk > 0.0) && fail == 0) fail = k + 1;
    const double inv = lm_rsqrt(pk > 0.0 ? pk : 1.0);
    uinv[k] = inv;
    ld += log(pk > 0.0 ? pk : 1.0);
    m[k] *= inv;
#pragma unroll
    for (int i = k + 1; i < LP; ++i) m[i] = __builtin_fma(-lm_rl(m[k], i), m[k], m[i]);
  }
#pragma unroll
  for (int i = 0; i < LP; ++i) asm volatile("" : "+v"(m[i]));
  double v[LP];
#pragma unroll
  for (int k = LP - 1; k >= 0; --k) {
    // (lanes LP ..: m holds this lane's forward-substituted right-hand side; the lanes that hold U itself must not move --
    //  what the unconditional elimination left below their diagonals would otherwise be fed back into U)
    v[k] = lane >= LP ? m[k] * uinv[k] : 0.0;
#pragma unroll
    for (int i = 0; i < k; ++i) m[i] -= lm_rl(m[i], k) * v[k];
  }
  if (fail == 0 && lane >= LP && lane < 2 * LP) {
    const int c = lane - LP;               // column c of the inverse; the upper triangle is written, and mirrored
#pragma unroll
    for (int k = 0; k < LP; ++k)
      if (k <= c) {
        cov[k * ldc + c] = v[k]; cov[c * ldc + k] = v[k];
        hcov[k * LP + c] = v[k]; hcov[c * LP + k] = v[k];
      }
  }
  if (lane == 0) *logdet = ld;             // log det = sum of log pivots (U_kk^2 = pivot k)
  return fail;
}

static __device__ __forceinline__ void lm_fit(LmArgs a) {
  __shared__ double sp[LP], spt[LP], sD[LP], sV[LP], sT[LP], sG[LP], sA[LP * LP], sq[LNQ], red[LRED], ss[16];
  __shared__ int si[4];
  __shared__ double sdata[LDATA];
  __shared__ double srow[LROWS * (LP + 1)];
  __shared__ int sblk[LROWS];
  const int tid = threadIdx.x;
  if (a.n_blocks && tid < LROWS) {   // row -> its covariance block (or none); the host sends such fits here with at most LROWS rows
    int b = -1;
    for (int q = 0; q < (int)a.n_blocks; ++q)
      if (tid >= a.blk_row0[q] && tid < a.blk_row0[q] + a.blk_size[q]) b = q;
    sblk[tid] = b;
  }
  // the data of a fit this small is read dozens of times: once from memory, then from LDS (the generated row loops read
  // through a.x / a.ymean / a.wdiag, which may as well point there)
  if (a.n_data * (LNX + 2) <= LDATA) {
    const int n = (int)a.n_data;
    for (int i = tid; i < n * LNX; i += 256) sdata[i] = a.x[i];
    for (int i = tid; i < n; i += 256) { sdata[n * LNX + i] = a.ymean[i]; sdata[n * (LNX + 1) + i] = a.wdiag[i]; }
    a.x = sdata; a.ymean = sdata + n * LNX; a.wdiag = sdata + n * (LNX + 1);
  }
  __shared__ double sprior[LP * LP + LP];
  if (a.prior_prec) {
    const int np = a.prior_dense ? LP * LP : LP;
    for (int i = tid; i < np; i += 256) sprior[i] = a.prior_prec[i];      // (up to 32 x 32 entries: more than one per thread)
    if (tid < LP) sprior[LP * LP + tid] = a.prior_mean[tid];
    a.prior_prec = sprior; a.prior_mean = sprior + LP * LP;
  }
  if (tid < LP) { sp[tid] = a.p0[tid]; sV[tid] = 0.0; spt[tid] = a.p0[tid]; }
  if (tid < 16) ss[tid] = 0.0;
  __syncthreads();
  const long long c_begin = clock64(), w_begin = wall_clock64();
  long long c_nrm = 0, c_solve = 0, c_res = 0;
  lm_normal(a, sp, red, sq, sA, sG, sT, ss, srow, sblk);
  c_nrm += clock64() - c_begin;
  int reason = 0, nit = 0, nfev = 1, njev = 1, ntrial = 0;
  if (!(ss[S_CHI2] - ss[S_CHI2] == 0.0)) reason = 2;
  if (tid == 0 && reason == 0) {
    // scale_init + the starting mu, nu, delta of do_init
    double mx = 0.0, dxn = 0.0;
    for (int j = 0; j < LP; ++j) {
      const double c2 = sA[j * LP + j], cn = sqrt(c2 > 0.0 ? c2 : 0.0);
      const double d = a.scaler == SC_LEVENBERG ? 1.0 : (cn == 0.0 ? 1.0 : cn);
      sD[j] = d;
      mx = fmax(mx, cn / d);
      dxn += d * sp[j] * d * sp[j];
    }
    ss[S_MU] = 1e-3 * mx * mx;
    ss[S_NU] = 2.0;
    ss[S_DELTA] = 0.3 * fmax(1.0, sqrt(dxn));
  }
  __syncthreads();
  while (reason == 0 && nit < a.maxit) {
    int bad_steps = 0;
    bool accepted = false;
    while (!accepted) {
      long long c0 = clock64();
      if (tid < 64) lm_solve(sA, sG, sD, sp, ss[S_MU], a.watch, sV, spt, ss, si);
      __syncthreads();
      c_solve += clock64() - c0;
      c0 = clock64();
      if (si[0] != 0 || ss[S_VFINITE] == 0.0 || ss[S_PIVMIN] < 1e-8) { reason = 2; break; }
      double ct = lm_res(a, spt, red, srow, sblk);
      c_res += clock64() - c0;
      if (a.prior_prec) {        // (every thread the same few terms, in the same order)
        double c = 0.0;
        for (int i = 0; i < LP; ++i) {
          double t;
          if (a.prior_dense) { t = 0.0; for (int k = 0; k < LP; ++k) t += a.prior_prec[i * LP + k] * (spt[k] - a.prior_mean[k]); }
          else t = a.prior_prec[i] * (spt[i] - a.prior_mean[i]);
          c += (spt[i] - a.prior_mean[i]) * t;
        }
        ct += c;
      }
      ++ntrial; ++nfev;
      if (tid == 0) {
        const double chi2 = ss[S_CHI2], mu = ss[S_MU];
        double rho = -1.0;
        if (ct < chi2) {         // (|f_t| < |f|; 1 - (|f_t| / |f|)^2 without the two square roots: the same to rounding.  NaN rejects)
          const double num = ss[S_VG] + mu * ss[S_DV2];        // = pred * chi2
          rho = num > 0.0 ? (chi2 - ct) / num : -1.0;
        }
        if (rho > 0.75) ss[S_DELTA] *= a.factor_up;
        else if (rho < 0.25) ss[S_DELTA] /= a.factor_down;
        if (rho > 0.0) {
          const double b = 2.0 * rho - 1.0;
          ss[S_MU] = mu * fmax(0.333333333333333, 1.0 - b * b * b);
          ss[S_NU] = 2.0;
        } else {
          ss[S_MU] = mu * ss[S_NU];
          ss[S_NU] *= 2.0;
        }
        ss[S_RHO] = rho;
        ss[S_CHI2_TRIAL] = ct;
        ss[S_ACCEPT] = rho > 0.0 ? 1.0 : 0.0;
        ss[S_SOLVED] = 1.0;
      }
      __syncthreads();
      accepted = ss[S_ACCEPT] != 0.0;
      if (!accepted && ++bad_steps > 15) break;
      __syncthreads();
    }
    if (reason) break;
    if (!accepted) {
      // sixteen rejections in a row: gsl_multifit_nlinear_driver tests convergence after an iteration without progress as
      // well, with the last (rejected) step as dx (api.hip iterate_device, lm_converge_kernel); at the very first iteration
      // that is the driver's early exit, left to the general path
      if (nit == 0) { reason = 2; break; }
      if (tid == 0) {
        double notx = 0.0, gn = 0.0;
        for (int j = 0; j < LP; ++j) {
          const double xj = sp[j];
          notx += (fabs(sV[j]) < a.xtol * a.xtol + a.xtol * fabs(xj)) ? 0.0 : 1.0;
          gn = fmax(gn, fabs(fmax(xj, 1.0) * sG[j]));
        }
        ss[S_INFO] = notx == 0.0 ? 1.0 : (gn <= a.gtol * fmax(0.5 * ss[S_CHI2], 1.0) ? 2.0 : 0.0);
      }
      __syncthreads();
      ++nit;
      if (ss[S_INFO] != 0.0) break;
      __syncthreads();
      continue;
    }
    if (tid < LP) sp[tid] = spt[tid];
    __syncthreads();
    { const long long c0 = clock64(); lm_normal(a, sp, red, sq, sA, sG, sT, ss, srow, sblk); c_nrm += clock64() - c0; }
    ++njev;
    if (!(ss[S_CHI2] - ss[S_CHI2] == 0.0)) { reason = 2; break; }
    if (tid < 64) {       // scaling update and gsl_multifit_nlinear_test, one lane per parameter
      double notx = 0.0, gn = 0.0;
      if (tid < LP) {
        const int j = tid;
        const double c2 = sA[j * LP + j], cn = sqrt(c2 > 0.0 ? c2 : 0.0);
        double d;
        if (a.scaler == SC_LEVENBERG) d = sD[j];
        else if (a.scaler == SC_MORE) d = fmax(sD[j], cn);
        else d = cn == 0.0 ? 1.0 : cn;
        sD[j] = d;
        const double xj = sp[j];
        notx = (fabs(sV[j]) < a.xtol * a.xtol + a.xtol * fabs(xj)) ? 0.0 : 1.0;
        gn = fabs(fmax(xj, 1.0) * sG[j]);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        notx += __shfl_xor(notx, o, 64);
        gn = fmax(gn, __shfl_xor(gn, o, 64));
      }
      if (tid == 0) ss[S_INFO] = notx == 0.0 ? 1.0 : (gn <= a.gtol * fmax(0.5 * ss[S_CHI2], 1.0) ? 2.0 : 0.0);
    }
    __syncthreads();
    ++nit;
    if (ss[S_INFO] != 0.0) break;
    __syncthreads();
  }
  if (reason == 0) reason = 1;
  __syncthreads();
  // the state where the general path keeps it (device), and mirrored for the host
  double *h = a.host;
  if (tid < LP) {
    a.p[tid] = sp[tid]; a.p_trial[tid] = spt[tid]; a.dscale[tid] = sD[tid]; a.v_out[tid] = sV[tid];
    a.coln2[tid] = sA[tid * LP + tid]; a.gvec[tid] = sG[tid];
    double *m = h + HMIR;
    m[tid] = sp[tid]; m[LP + 1 + tid] = sG[tid]; m[2 * (LP + 1) + tid] = sD[tid]; m[3 * (LP + 1) + tid] = sA[tid * LP + tid];
    m[4 * (LP + 1) + tid] = sV[tid];
  }
  for (int e = tid; e < LP * LP; e += 256) a.apk[(e / LP) * 128 + (e % LP)] = sA[e];
  if (tid == 0) {
    a.gvec[LP] = ss[S_CHI2];
    h[HMIR + LP + 1 + LP] = ss[S_CHI2];
    ss[S_SEQ] = 0.0;
    ss[S_HOSTPTR] = a.hostptr_bits;
  }
  __syncthreads();
  if (tid < 16) { a.st[tid] = ss[tid]; h[tid] = ss[tid]; }
  if (tid < 64 && reason == 1 && a.want_cov) {
    const int bad = lm_cov(sA, a.cov, a.ldc, h + HCOV, h + 22);
    if (tid == 0) h[21] = bad == 0 ? 1.0 : 0.0;
  } else if (tid == 0) h[21] = 0.0;
  if (tid == 0) {
    h[16] = (double)reason; h[17] = nit; h[18] = nfev; h[19] = njev; h[20] = ntrial; h[23] = 0.0;
    // diagnostics: shader cycles in all / in the normal equations / in the solves / in the trial residuals, 100 MHz ticks in all
    m_diag(h + HDIAG, clock64() - c_begin, c_nrm, c_solve, c_res, wall_clock64() - w_begin);
  }
  if (a.pub) {
    // Hand-off to a host that POLLS pinned memory instead of waiting for the stream.  Stores to host memory reach it in no
    // particular order -- a system-scope fence between two of them does not make the first visible to the CPU before the
    // second (measured: the flag was seen with the covariance words of the PREVIOUS fit still in place, 3 times in 3600
    // fits) -- so nothing here relies on order.  The finished block (device memory, above) is copied out word by word with a
    // position-weighted sum over the words; the sum (seeded with this launch's sequence number) goes to word 23 and the flag
    // word 16 = reason | info << 8 | nit << 16 | seq << 40 is what the host waits for.  The host takes a snapshot, recomputes
    // the sum and believes the snapshot only when both agree (api.hip run_one_launch); otherwise it keeps polling, then
    // falls back to hipStreamSynchronize + a copy of the device block.
    __threadfence();
    __syncthreads();
    constexpr int NW = HCOV + LP * LP;
    const unsigned long long *hw = reinterpret_cast<const unsigned long long *>(h);
    unsigned long long acc = 0;
    for (int i = tid; i < NW; i += 256) {
      if (i == 16 || i == 23) continue;
      const unsigned long long w = hw[i];
      a.pub[i] = w;
      acc += (w ^ 0x9E3779B97F4A7C15ull) * (2ull * (unsigned long long)i + 1ull);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)(acc & 0xffffffffull), m, 64), hi = __shfl_xor((unsigned)(acc >> 32), m, 64);
      acc += ((unsigned long long)hi << 32) | lo;
    }
    unsigned long long *spub = reinterpret_cast<unsigned long long *>(red);
    if ((tid & 63) == 0) spub[tid >> 6] = acc;
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      a.pub[23] = spub[0] + spub[1] + spub[2] + spub[3] + a.seq * 0xD6E8FEB86659FD93ull;
      __threadfence_system();
      a.pub[16] = (unsigned long long)(reason & 0xff) | ((unsigned long long)((int)ss[S_INFO] & 0xff) << 8) |
                  ((unsigned long long)(nit & 0xffffff) << 16) | (a.seq << 40);
      __threadfence_system();
    }
  }
}

// ===SINGLE===
extern "C" __global__ __launch_bounds__(256) void lsqamd_jit_lm(LmArgs a) { lm_fit(a); }
// ===BATCH===

// Many same-shape fits (bootstrap / simulated copies, a sweep of priors): one workgroup per fit, the same loop.  Per-fit
// inputs and outputs are strided; a fit's record (the block the single-fit kernel mirrors to the host) goes to device
// scratch and its scalars into the batched engine's per-fit arrays.  reason: 1 done (covariance formed), 3 done (not
// formed), 2 irregular -- the host then runs the lockstep engine.
struct LmBatch {
  long long ymean_stride, prec_stride, tile_stride, cov_stride, scratch_stride;
  double *logdet, *mu, *chi2;
  int *nit, *info, *status, *nfev, *njev, *active, *reason;
};

extern "C" __global__ __launch_bounds__(256) void lsqamd_jit_lmb(LmArgs a, LmBatch b) {
  const long long fit = blockIdx.x;
  a.ymean += fit * b.ymean_stride;
  a.p0 += fit * LP; a.p += fit * LP; a.p_trial += fit * LP; a.dscale += fit * LP; a.v_out += fit * LP;
  a.apk += fit * b.tile_stride; a.gvec += fit * b.tile_stride;
  if (a.prior_prec) { a.prior_mean += fit * LP; a.prior_prec += fit * b.prec_stride; }
  a.cov += fit * b.cov_stride;
  a.host += fit * b.scratch_stride;          // [0, HREC) the record block, then the LM record (16) and the column norms (LP)
  a.st = a.host + HREC;
  a.coln2 = a.host + HREC + 16;
  lm_fit(a);
  __syncthreads();
  if (threadIdx.x == 0) {
    const double *h = a.host;
    const int reason = (int)h[16];
    b.mu[fit] = h[S_MU]; b.chi2[fit] = h[S_CHI2];
    b.nit[fit] = (int)h[17]; b.nfev[fit] = (int)h[18]; b.njev[fit] = (int)h[19];
    b.info[fit] = (int)h[S_INFO];
    b.status[fit] = h[S_INFO] != 0.0 ? 0 : -2;
    b.logdet[fit] = h[22];
    b.active[fit] = 0;
    b.reason[fit] = reason == 1 ? (h[21] == 1.0 ? 1 : 3) : 2;
  }
}

)LSQLM";

// batch_only: the module with the batched whole-fit kernel alone (built the first time a batch asks for it)
std::string generate(const Plan &pl, bool batch_only = false) {
  Src o;
  o.s += "// generated by lsqfit_amd (jit.hip) from an expression tape\n";
  o.s += kDevmath;
  o.s += "\nusing namespace lsqamd;\n";
  o.s += "struct Args { const double *x, *p, *ymean, *wdiag; const unsigned char *in_block; double *out_w, *out_raw; long long ld, n_data;\n"
         "              long long p_stride, out_stride, ymean_stride; const int *batch_active; };\n";
  o.s += "static __device__ __forceinline__ double wsum(double v) {\n"
         "#pragma unroll\n  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);\n  return v;\n}\n";
  // index tables of the non-affine private slots
  int gid = 0;
  for (const WSum &w : pl.wsums)
    for (const Group &g : w.groups) {
      for (int s = 0; s < g.n_slots; ++s)
        if (!g.shared[(size_t)s] && g.stride[(size_t)s] == INT32_MIN) {
          o.f("static __device__ const int T%d_%d[%d] = {", gid, s, g.n_terms);
          for (int k = 0; k < g.n_terms; ++k) o.f("%d,", g.idx[(size_t)s][(size_t)k]);
          o.s += "};\n";
        }
      for (size_t c = 0; c < g.cvals.size(); ++c)
        if (g.cvals[c].size() > 1) {     // a constant that differs from term to term
          o.f("static __device__ const double CT%d_%zu[%d] = {", gid, c, g.n_terms);
          for (int k = 0; k < g.n_terms; ++k) o.f("%s,", dlit(g.cvals[c][(size_t)k]).c_str());
          o.s += "};\n";
        }
      ++gid;
    }
  if (!pl.unread.empty()) {
    o.f("static __device__ const int ZC[%zu] = {", pl.unread.size());
    for (int j : pl.unread) o.f("%d,", j);
    o.s += "};\n";
  }
  std::vector<char> xused((size_t)pl.n_x, 0);
  for (const Node &nd : pl.nodes)
    if (nd.op == LSQAMD_OP_X) xused[(size_t)nd.arg] = 1;
  const int nout = (int)pl.out_params.size();
  for (int jac = 0; jac < (batch_only ? 0 : 2); ++jac) {
    o.f("extern \"C\" __global__ __launch_bounds__(256) void %s(Args a) {\n", jac ? "lsqamd_jit_jac" : "lsqamd_jit_res");
    // blockIdx.y = fit of a batch (lockstep fits, chi2 at many points): its parameters, data means and output rows
    o.s += "  if (a.batch_active && !a.batch_active[blockIdx.y]) return;\n";
    o.s += "  a.p += (long long)blockIdx.y * a.p_stride;\n  a.ymean += (long long)blockIdx.y * a.ymean_stride;\n";
    o.s += "  a.out_w += (long long)blockIdx.y * a.out_stride;\n  if (a.out_raw) a.out_raw += (long long)blockIdx.y * a.out_stride;\n";
    o.f("  __shared__ double sp[%d];\n", pl.P < 1 ? 1 : pl.P);
    o.f("  for (int i = threadIdx.x; i < %d; i += 256) sp[i] = a.p[i];\n  __syncthreads();\n", pl.P);
    if (pl.wave_per_row) {
      o.s += "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n";
      o.s += "  for (long long row = (long long)blockIdx.x * 4 + wave; row < a.n_data; row += (long long)gridDim.x * 4) {\n";
    } else {
      o.s += "  for (long long row = (long long)blockIdx.x * 256 + threadIdx.x; row < a.n_data; row += (long long)gridDim.x * 256) {\n";
    }
    for (int i = 0; i < pl.n_x; ++i)
      if (xused[(size_t)i]) o.f("    const double x%d = a.x[row * %d + %d];\n", i, pl.n_x, i);
    o.s += "    const bool blk = a.in_block && a.in_block[row];\n    const double w = blk ? 1.0 : a.wdiag[row];\n";
    if (jac) o.s += "    double *dst = (blk ? a.out_raw : a.out_w) + row * a.ld;\n";
    if (jac)
      for (int j = 0; j < nout; ++j) o.f("    double oacc%d = 0.0, sh%d = 0.0;\n", j, j);
    // phase 1: the wide sums
    gid = 0;
    for (size_t wi = 0; wi < pl.wsums.size(); ++wi) {
      const WSum &w = pl.wsums[wi];
      o.f("    double S%zu = 0.0;\n", wi);
      for (const Group &g : w.groups) {
        o.s += "    {\n      double acc = 0.0;\n";
        const int mode = jac && w.fused ? 1 : 0;
        gen_group(pl, o, g, gid, mode, w.fused > 0 ? "1.0" : "-1.0", "acc");
        o.f("      S%zu %s wsum(acc);\n    }\n", wi, g.sign > 0 ? "+=" : "-=");
        ++gid;
      }
    }
    // phase 2: the outer expression, forward and (jac) reverse
    {
      TreeGen tg(pl.outer, pl, o, "o", "    ", jac != 0);
      tg.leaf_value = [&](const Node &nd) {
        if (nd.op == OP_WSUM) return "S" + std::to_string(nd.arg);
        return "sp[" + std::to_string(nd.arg) + "]";
      };
      if (jac)
        for (size_t wi = 0; wi < pl.wsums.size(); ++wi)
          if (!pl.wsums[wi].fused) o.f("    double aS%zu = 0.0;\n", wi);
      tg.leaf_adjoint = [&](const Node &nd, const std::string &gg) {
        if (nd.op == OP_WSUM) {
          if (!pl.wsums[(size_t)nd.arg].fused) o.f("    aS%d += %s;\n", nd.arg, gg.c_str());
          return;
        }
        int j = 0;
        while (pl.out_params[(size_t)j] != nd.arg) ++j;
        o.f("    oacc%d += %s;\n", j, gg.c_str());
      };
      tg.forward(pl.oroot);
      o.f("    const double fval = %s;\n", tg.v(pl.oroot).c_str());
      if (jac) {
        o.s += "    const double one = 1.0;\n";
        tg.reverse(pl.oroot, "one");
      }
    }
    if (jac) {
      // phase 3: sums whose adjoint came out of the outer sweep
      gid = 0;
      for (size_t wi = 0; wi < pl.wsums.size(); ++wi) {
        const WSum &w = pl.wsums[wi];
        for (const Group &g : w.groups) {
          if (!w.fused) {
            o.s += "    {\n";
            gen_group(pl, o, g, gid, 2, "aS" + std::to_string(wi), "");
            o.s += "    }\n";
          }
          ++gid;
        }
      }
      if (!pl.unread.empty()) {   // columns of parameters the formula does not read
        if (pl.wave_per_row) o.f("    for (int k = lane; k < %zu; k += 64) dst[ZC[k]] = 0.0;\n", pl.unread.size());
        else o.f("    for (int k = 0; k < %zu; ++k) dst[ZC[k]] = 0.0;\n", pl.unread.size());
      }
      if (pl.wave_per_row) {
        if (nout > 0) {
          bool any_shared = false;
          for (const WSum &w : pl.wsums)
            for (const Group &g : w.groups)
              for (int s = 0; s < g.n_slots; ++s) any_shared = any_shared || g.shared[(size_t)s];
          o.s += "    double mine = 0.0; int mycol = 0;\n";
          for (int j = 0; j < nout; ++j) {
            bool sh = false;
            for (const WSum &w : pl.wsums)
              for (const Group &g : w.groups)
                for (int s = 0; s < g.n_slots; ++s) sh = sh || (g.shared[(size_t)s] && g.base[(size_t)s] == pl.out_params[(size_t)j]);
            if (sh) o.f("    oacc%d += wsum(sh%d);\n", j, j);
            o.f("    if (lane == %d) { mine = oacc%d; mycol = %d; }\n", j, j, pl.out_params[(size_t)j]);
          }
          (void)any_shared;
          o.f("    if (lane < %d) dst[mycol] = w * mine;\n", nout);
        }
        o.f("    if (lane == 63) dst[%d] = w * (fval - a.ymean[row]);\n", pl.P);
      } else {
        for (int j = 0; j < nout; ++j) o.f("    dst[%d] = w * oacc%d;\n", pl.out_params[(size_t)j], j);
        o.f("    dst[%d] = w * (fval - a.ymean[row]);\n", pl.P);
      }
    } else {
      if (pl.wave_per_row) o.s += "    if (lane == 0) (blk ? a.out_raw : a.out_w)[row] = w * (fval - a.ymean[row]);\n";
      else o.s += "    (blk ? a.out_raw : a.out_w)[row] = w * (fval - a.ymean[row]);\n";
    }
    o.s += "  }\n}\n";
  }
  if (pl.nrm_ok && !batch_only) {
    // ---- normal equations without the Jacobian: per lane P (P + 1) / 2 + P + 1 running sums over its rows
    const int P = pl.P, NA = P * (P + 1) / 2, NQ = NA + P + 1;
    o.f("extern \"C\" __global__ __launch_bounds__(256) void lsqamd_jit_nrm(Args a) {\n");
    o.f("  __shared__ double sp[%d];\n  __shared__ double red[4][%d];\n", P, NQ);
    o.f("  for (int i = threadIdx.x; i < %d; i += 256) sp[i] = a.p[i];\n  __syncthreads();\n", P);
    o.s += "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n";
    for (int i = 0; i < P; ++i)
      for (int j = i; j < P; ++j) o.f("  double nA%d_%d = 0.0;\n", i, j);
    for (int i = 0; i < P; ++i) o.f("  double nG%d = 0.0;\n", i);
    o.s += "  double nC = 0.0;\n";
    o.s += "  for (long long row = (long long)blockIdx.x * 256 + threadIdx.x; row < a.n_data; row += (long long)gridDim.x * 256) {\n";
    for (int i = 0; i < pl.n_x; ++i)
      if (xused[(size_t)i]) o.f("    const double x%d = a.x[row * %d + %d];\n", i, pl.n_x, i);
    o.s += "    const double w = a.wdiag[row];\n";
    for (int j = 0; j < nout; ++j) o.f("    double oacc%d = 0.0;\n", j);
    {
      TreeGen tg(pl.outer, pl, o, "o", "    ", true);
      tg.leaf_value = [&](const Node &nd) { return "sp[" + std::to_string(nd.arg) + "]"; };
      tg.leaf_adjoint = [&](const Node &nd, const std::string &gg) {
        int j = 0;
        while (pl.out_params[(size_t)j] != nd.arg) ++j;
        o.f("    oacc%d += %s;\n", j, gg.c_str());
      };
      tg.forward(pl.oroot);
      o.f("    const double fval = %s;\n    const double one = 1.0;\n", tg.v(pl.oroot).c_str());
      tg.reverse(pl.oroot, "one");
    }
    o.s += "    const double rr = w * (fval - a.ymean[row]);\n";
    std::vector<int> slot((size_t)P, -1);
    for (int j = 0; j < nout; ++j) slot[(size_t)pl.out_params[(size_t)j]] = j;
    for (int i = 0; i < P; ++i) {
      if (slot[(size_t)i] >= 0) o.f("    const double dd%d = w * oacc%d;\n", i, slot[(size_t)i]);
      else o.f("    const double dd%d = 0.0;\n", i);
    }
    for (int i = 0; i < P; ++i)
      for (int j = i; j < P; ++j) o.f("    nA%d_%d += dd%d * dd%d;\n", i, j, i, j);
    for (int i = 0; i < P; ++i) o.f("    nG%d += dd%d * rr;\n", i, i);
    o.s += "    nC += rr * rr;\n  }\n";
    int q = 0;
    for (int i = 0; i < P; ++i)
      for (int j = i; j < P; ++j) { o.f("  { const double t = wsum(nA%d_%d); if (lane == 0) red[wave][%d] = t; }\n", i, j, q); ++q; }
    for (int i = 0; i < P; ++i) { o.f("  { const double t = wsum(nG%d); if (lane == 0) red[wave][%d] = t; }\n", i, q); ++q; }
    o.f("  { const double t = wsum(nC); if (lane == 0) red[wave][%d] = t; }\n", q);
    o.f("  __syncthreads();\n  if (threadIdx.x < %d) a.out_w[(long long)blockIdx.x * %d + threadIdx.x] = "
        "red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];\n}\n", NQ, NQ);
  }
  if (pl.fit_ok) {
    // ---- the same sums and the residual as functions of ONE workgroup, and the whole-fit kernels over them.  Two forms of the
    // normal equations: up to NRM_MAX_P parameters every thread keeps all the sums of its rows in registers; beyond that (up to
    // FIT_MAX_P) the rows go through LDS, fit_wide_rows(P) at a time, and thread q adds up product q of the chunk.
    const int P = pl.P, NA = P * (P + 1) / 2, NQ = NA + P + 1;
    const bool regs = P <= lsqamd_jit::NRM_MAX_P;
    const int lrows = regs ? lsqamd_jit::FIT_MAX_BLOCK_ROWS : lsqamd_jit::fit_wide_rows(P), lred = regs ? 16 * NQ : 16;
    // LDS budget (64 KB of static LDS per workgroup): the rows of a correlated (or wide) fit, P + 1 values each, the row totals
    // of the register form, A, the prior, ~2 KB of small arrays -- the rest stages x, y, w
    const int lds_fixed = 8 * (lrows * (P + 1) + lred + 2 * P * P + 8 * P + NQ + 32) + 4 * lrows + 768;
    int ldata = (64512 - lds_fixed) / 8;
    if (ldata > 12288) ldata = 12288;
    if (ldata < 16) ldata = 16;
    o.f("constexpr int LP = %d, LNA = %d, LNQ = %d, LNX = %d, LDATA = %d, LROWS = %d, LRED = %d;\n", P, NA, NQ, pl.n_x < 1 ? 1 : pl.n_x,
        ldata, lrows, lred);
    o.f("constexpr int HMIR = 24, HDIAG = %d, HCOV = %d, HREC = %d;      // the record block: mirrors, cycle counters, covariance, size\n",
        24 + 5 * (P + 1), 24 + 5 * (P + 1) + 8, (24 + 5 * (P + 1) + 8 + P * P + 15) / 16 * 16);
    if (!regs) {   // which two columns of a row product q multiplies: A upper row-major, then J^T f, then |f|^2
      std::string qi = "static __device__ const unsigned char QI[LNQ] = {", qj = "static __device__ const unsigned char QJ[LNQ] = {";
      for (int i = 0; i < P; ++i)
        for (int j = i; j < P; ++j) { qi += std::to_string(i) + ","; qj += std::to_string(j) + ","; }
      for (int i = 0; i < P; ++i) { qi += std::to_string(i) + ","; qj += std::to_string(P) + ","; }
      qi += std::to_string(P) + "};\n"; qj += std::to_string(P) + "};\n";
      o.s += qi + qj;
    }
    o.f("enum { S_CHI2 = %d, S_MU = %d, S_NU = %d, S_DELTA = %d, S_VG = %d, S_DV2 = %d, S_VFINITE = %d, S_RHO = %d, S_CHI2_TRIAL = %d, "
        "S_ACCEPT = %d, S_SOLVED = %d, S_INFO = %d, S_PIVMIN = %d, S_SEQ = %d, S_HOSTPTR = %d, SC_LEVENBERG = %d, SC_MORE = %d };\n",
        (int)lsqamd::LMS_CHI2, (int)lsqamd::LMS_MU, (int)lsqamd::LMS_NU, (int)lsqamd::LMS_DELTA, (int)lsqamd::LMS_VG, (int)lsqamd::LMS_DV2, (int)lsqamd::LMS_VFINITE, (int)lsqamd::LMS_RHO,
        (int)lsqamd::LMS_CHI2_TRIAL, (int)lsqamd::LMS_ACCEPT, (int)lsqamd::LMS_SOLVED, (int)lsqamd::LMS_INFO, (int)lsqamd::LMS_PIVMIN, (int)lsqamd::LMS_SEQ, (int)lsqamd::LMS_HOSTPTR,
        (int)LSQAMD_SCALE_LEVENBERG, (int)LSQAMD_SCALE_MORE);
    // wave sum without LDS round trips: four DPP steps inside each row of 16 lanes (every lane of a row ends up with the
    // row's total), then the four row totals through scalar registers
    o.s += "static __device__ __forceinline__ double lm_rl(double v, int lane) {\n"
           "  union { double d; int i[2]; } u;\n  u.d = v;\n  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);\n"
           "  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);\n  return u.d;\n}\n"
           "template <int CTRL> static __device__ __forceinline__ double lm_dpp(double v) {\n"
           "  union { double d; int i[2]; } u;\n  u.d = v;\n  u.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, false);\n"
           "  u.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, false);\n  return u.d;\n}\n"
           "static __device__ __forceinline__ double lm_rowsum(double v) {\n"
           "  v += lm_dpp<0xB1>(v);      // quad_perm [1,0,3,2]\n  v += lm_dpp<0x4E>(v);      // quad_perm [2,3,0,1]\n"
           "  v += lm_dpp<0x141>(v);     // row_half_mirror\n  v += lm_dpp<0x140>(v);     // row_mirror\n  return v;\n}\n"
           "static __device__ __forceinline__ double lm_wsum(double v) {\n"
           "  v = lm_rowsum(v);\n  return (lm_rl(v, 0) + lm_rl(v, 16)) + (lm_rl(v, 32) + lm_rl(v, 48));\n}\n";
    const char *decl = strstr(kLmDriver, "struct LmArgs {");
    const char *decl_end = strstr(decl, "};");
    o.s.append(decl, (size_t)(decl_end - decl) + 2);
    o.s += "\n";
    // correlated rows (a.n_blocks > 0, at most LROWS rows, one per thread): row m of block b comes out as
    // sum_k Wt_b[k][m] raw[row0_b + k] -- what block_whiten_vec_kernel and the whitening GEMM compute -- from the raw rows
    // the workgroup has just filed in LDS; rows outside blocks pass through (they were weighted when they were formed)
    o.s += "template <int NC> static __device__ __forceinline__ void lm_whiten_row(const LmArgs &a, const double *srow, const int *sblk, int row, double *o) {\n"
           "  const int b = sblk[row];\n"
           "  if (b < 0) {\n#pragma unroll\n    for (int c = 0; c < NC; ++c) o[c] = srow[row * NC + c];\n    return;\n  }\n"
           "  const int B = (int)a.blk_size[b], r0 = (int)a.blk_row0[b], m = row - r0;\n"
           "  const double *W = a.wt + a.blk_woff[b] + m;\n"
           "#pragma unroll\n  for (int c = 0; c < NC; ++c) o[c] = 0.0;\n"
           "  if (NC <= 8) {      // few columns: the compiler's own eight-fold unrolling keeps the loads ahead of their use\n"
           "#pragma unroll 8\n    for (int k = 0; k < B; ++k) {\n      const double wv = W[(long long)k * B];\n      const double *sr = srow + (r0 + k) * NC;\n"
           "#pragma unroll\n      for (int c = 0; c < NC; ++c) o[c] = __builtin_fma(wv, sr[c], o[c]);\n    }\n    return;\n  }\n"
           "  int k0 = 0;\n  for (; k0 + 8 <= B; k0 += 8) {      // many columns: eight W^T rows requested explicitly before the first is used\n"
           "    double wv[8];\n#pragma unroll\n    for (int u = 0; u < 8; ++u) wv[u] = W[(long long)(k0 + u) * B];\n"
           "#pragma unroll\n    for (int u = 0; u < 8; ++u) {\n      const double *sr = srow + (r0 + k0 + u) * NC;\n"
           "#pragma unroll\n      for (int c = 0; c < NC; ++c) o[c] = __builtin_fma(wv[u], sr[c], o[c]);\n    }\n  }\n"
           "  for (; k0 < B; ++k0) {\n    const double wv = W[(long long)k0 * B];\n    const double *sr = srow + (r0 + k0) * NC;\n"
           "#pragma unroll\n    for (int c = 0; c < NC; ++c) o[c] = __builtin_fma(wv, sr[c], o[c]);\n  }\n}\n";
    for (int fn = 0; fn < 2; ++fn) {
      const bool nrm = fn == 0;
      if (nrm) o.s += "static __device__ void lm_nrm(const LmArgs &a, const double *sp, double *red, double *sq, double *srow, const int *sblk) {\n";
      else o.s += "static __device__ double lm_res(const LmArgs &a, const double *sp, double *red, double *srow, const int *sblk) {\n";
      o.s += "  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n";
      if (nrm && regs) {
        for (int i = 0; i < P; ++i)
          for (int j = i; j < P; ++j) o.f("  double nA%d_%d = 0.0;\n", i, j);
        for (int i = 0; i < P; ++i) o.f("  double nG%d = 0.0;\n", i);
      }
      o.s += "  double nC = 0.0;\n";
      if (nrm && !regs) {
        // wide form: LROWS rows at a time through LDS (the products of a chunk are added to the running sums of thread q)
        for (int k = 0; k < (NQ + 255) / 256; ++k) o.f("  double qa%d = 0.0;\n", k);
        o.s += "  for (long long base = 0; base < a.n_data; base += LROWS) {\n  const long long row = base + threadIdx.x;\n"
               "  if (threadIdx.x < LROWS && row < a.n_data) {\n";
      } else {
        o.s += "  for (long long row = threadIdx.x; row < a.n_data; row += 256) {\n";
      }
      for (int i = 0; i < pl.n_x; ++i)
        if (xused[(size_t)i]) o.f("    const double x%d = a.x[row * %d + %d];\n", i, pl.n_x, i);
      o.s += "    const double w = (a.n_blocks && a.in_block[row]) ? 1.0 : a.wdiag[row];\n";
      if (nrm)
        for (int j = 0; j < nout; ++j) o.f("    double oacc%d = 0.0;\n", j);
      {
        TreeGen tg(pl.outer, pl, o, "o", "    ", nrm);
        tg.leaf_value = [&](const Node &nd) { return "sp[" + std::to_string(nd.arg) + "]"; };
        tg.leaf_adjoint = [&](const Node &nd, const std::string &gg) {
          int j = 0;
          while (pl.out_params[(size_t)j] != nd.arg) ++j;
          o.f("    oacc%d += %s;\n", j, gg.c_str());
        };
        tg.forward(pl.oroot);
        o.f("    const double fval = %s;\n", tg.v(pl.oroot).c_str());
        if (nrm) {
          o.s += "    const double one = 1.0;\n";
          tg.reverse(pl.oroot, "one");
        }
      }
      o.s += "    const double rr = w * (fval - a.ymean[row]);\n";
      if (nrm) {
        std::vector<int> slot((size_t)P, -1);
        for (int j = 0; j < nout; ++j) slot[(size_t)pl.out_params[(size_t)j]] = j;
        for (int i = 0; i < P; ++i) {
          if (slot[(size_t)i] >= 0) o.f("    const double dd%d = w * oacc%d;\n", i, slot[(size_t)i]);
          else o.f("    const double dd%d = 0.0;\n", i);
        }
        if (!regs) {
          // wide form: the row goes to LDS; whitened in place (every thread forms its row from the raw ones, then all write:
          // correlated fits come with at most LROWS rows, i.e. one chunk)
          o.s += "    double *sr = srow + threadIdx.x * (LP + 1);\n";
          for (int i = 0; i < P; ++i) o.f("    sr[%d] = dd%d;\n", i, i);
          o.s += "    sr[LP] = rr;\n  }\n  __syncthreads();\n";
          o.s += "  if (a.n_blocks) {\n    double o[LP + 1];\n    const bool mine = threadIdx.x < a.n_data && sblk[threadIdx.x] >= 0;\n"
                 "    if (mine) lm_whiten_row<LP + 1>(a, srow, sblk, threadIdx.x, o);\n    __syncthreads();\n"
                 "    if (mine) {\n#pragma unroll\n      for (int c = 0; c <= LP; ++c) srow[threadIdx.x * (LP + 1) + c] = o[c];\n    }\n"
                 "    __syncthreads();\n  }\n";
          o.s += "  const int nr = (int)(a.n_data - base < LROWS ? a.n_data - base : LROWS);\n";
          for (int k = 0; k < (NQ + 255) / 256; ++k)
            o.f("  if (threadIdx.x + %d < LNQ) {\n    const int ci = QI[threadIdx.x + %d], cj = QJ[threadIdx.x + %d];\n    double t = qa%d;\n"
                "#pragma unroll 8\n    for (int r = 0; r < nr; ++r) t = __builtin_fma(srow[r * (LP + 1) + ci], srow[r * (LP + 1) + cj], t);\n    qa%d = t;\n  }\n",
                256 * k, 256 * k, 256 * k, k, k);
          o.s += "  __syncthreads();\n  }\n";
          for (int k = 0; k < (NQ + 255) / 256; ++k) o.f("  if (threadIdx.x + %d < LNQ) sq[threadIdx.x + %d] = qa%d;\n", 256 * k, 256 * k, k);
          o.s += "  __syncthreads();\n  (void)lane; (void)wave; (void)nC; (void)red;\n}\n";
          continue;
        }
        o.s += "    if (!a.n_blocks) {\n";
        for (int i = 0; i < P; ++i)
          for (int j = i; j < P; ++j) o.f("      nA%d_%d += dd%d * dd%d;\n", i, j, i, j);
        for (int i = 0; i < P; ++i) o.f("      nG%d += dd%d * rr;\n", i, i);
        o.s += "      nC += rr * rr;\n    } else {\n      double *sr = srow + row * (LP + 1);\n";
        for (int i = 0; i < P; ++i) o.f("      sr[%d] = dd%d;\n", i, i);
        o.s += "      sr[LP] = rr;\n    }\n  }\n";
        o.s += "  if (a.n_blocks) {\n    __syncthreads();\n    if (threadIdx.x < a.n_data) {\n      double o[LP + 1];\n"
               "      lm_whiten_row<LP + 1>(a, srow, sblk, threadIdx.x, o);\n";
        for (int i = 0; i < P; ++i)
          for (int j = i; j < P; ++j) o.f("      nA%d_%d += o[%d] * o[%d];\n", i, j, i, j);
        for (int i = 0; i < P; ++i) o.f("      nG%d += o[%d] * o[LP];\n", i, i);
        o.s += "      nC += o[LP] * o[LP];\n    }\n  }\n";
      } else {
        o.s += "    if (!a.n_blocks) nC += rr * rr;\n    else srow[row] = rr;\n  }\n";
        o.s += "  if (a.n_blocks) {\n    __syncthreads();\n    if (threadIdx.x < a.n_data) {\n      double o[1];\n"
               "      lm_whiten_row<1>(a, srow, sblk, threadIdx.x, o);\n      nC += o[0] * o[0];\n    }\n  }\n";
      }
      if (nrm) {
        // all the wave sums first, as straight-line code (independent butterflies the scheduler interleaves: with a store under
        // `if (lane == 0)` after each, every one of them waited out its six cross-lane round trips alone), then the stores
        int qq = 0;
        for (int i = 0; i < P; ++i)
          for (int j = i; j < P; ++j) { o.f("  const double t%d = lm_rowsum(nA%d_%d);\n", qq, i, j); ++qq; }
        for (int i = 0; i < P; ++i) { o.f("  const double t%d = lm_rowsum(nG%d);\n", qq, i); ++qq; }
        o.f("  const double t%d = lm_rowsum(nC);\n", qq);
        // (every lane of a row of 16 holds the row's totals: its first lane files them, then thread q adds the 16 rows' q-th)
        o.s += "  if ((lane & 15) == 0) {\n    double *r = red + (threadIdx.x >> 4) * LNQ;\n";
        for (int q2 = 0; q2 <= qq; ++q2) o.f("    r[%d] = t%d;\n", q2, q2);
        o.s += "  }\n";
        o.s += "  __syncthreads();\n  if (threadIdx.x < LNQ) {\n    double t = red[threadIdx.x];\n#pragma unroll\n"
               "    for (int r = 1; r < 16; ++r) t += red[r * LNQ + threadIdx.x];\n    sq[threadIdx.x] = t;\n  }\n  __syncthreads();\n}\n";
      } else {
        o.s += "  { const double t = lm_wsum(nC); if (lane == 0) red[wave] = t; }\n  __syncthreads();\n"
               "  const double tot = red[0] + red[1] + red[2] + red[3];\n  __syncthreads();\n  return tot;\n}\n";
      }
    }
    {   // the driver: lm_normal, lm_solve, lm_fit, then ONE of the two kernels over it
      const char *rest = decl_end + 2, *single = strstr(rest, "// ===SINGLE==="), *batch = strstr(rest, "// ===BATCH===");
      o.s.append(rest, (size_t)(single - rest));
      if (batch_only) o.s += batch;
      else o.s.append(single, (size_t)(batch - single));
    }
  }
  return o.s;
}

// ---- build, cache, load ----------------------------------------------------------------------------
uint64_t fnv1a(const std::string &s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

// The on-disk cache of code objects.  A directory that is not ours alone is not used: its files are loaded and RUN on the GPU.
std::string cache_dir() {
  const char *e = getenv("LSQAMD_JIT_CACHE");
  std::string d;
  if (e && *e) d = e;
  else if ((e = getenv("XDG_CACHE_HOME")) && *e) d = std::string(e) + "/lsqfit_amd";
  else if ((e = getenv("HOME")) && *e) d = std::string(e) + "/.cache/lsqfit_amd";
  else d = "/tmp/lsqfit_amd_jit_" + std::to_string((long)getuid());
  if (d == "off" || d == "0") return "";
  std::string acc;
  for (size_t i = 0; i <= d.size(); ++i) {   // mkdir -p
    if (i == d.size() || (d[i] == '/' && i > 0)) {
      acc = d.substr(0, i);
      (void)mkdir(acc.c_str(), 0700);
    }
  }
  struct stat sb;
  if (lstat(d.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & (S_IWGRP | S_IWOTH)))
    return "";      // missing, a link, someone else's, or writable by others: no disk cache
  return d;
}

// cache file = header {magic, byte count, FNV-1a of the bytes} + the code object: a truncated or damaged file is recognised
// (and removed) here, not by a failing hipModuleLoadData that would leave the formula on the interpreter for good
constexpr uint64_t CACHE_MAGIC = 0x314a444d4151534cull;   // "LSQAMDJ1"
uint64_t fnv1a_bytes(const char *p, size_t n, uint64_t h = 1469598103934665603ull) {
  for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
  return h;
}

std::string cache_file(const std::string &src) {
  Rtc &r = rtc();
  char name[64];
  snprintf(name, sizeof(name), "%016" PRIx64 "%016" PRIx64, fnv1a(src), fnv1a(src + std::to_string(r.vmajor * 1000 + r.vminor), 88172645463325252ull));
  const std::string dir = cache_dir();
  return dir.empty() ? "" : dir + "/" + name + ".hsaco";
}

bool cache_read(const std::string &file, std::vector<char> &code) {
  FILE *fh = fopen(file.c_str(), "rb");
  if (!fh) return false;
  uint64_t head[3] = {0, 0, 0};
  bool ok = fread(head, sizeof(uint64_t), 3, fh) == 3 && head[0] == CACHE_MAGIC && head[1] > 0 && head[1] < (1ull << 31);
  if (ok) {
    code.resize((size_t)head[1]);
    ok = fread(code.data(), 1, code.size(), fh) == code.size() && fgetc(fh) == EOF && fnv1a_bytes(code.data(), code.size()) == head[2];
  }
  fclose(fh);
  if (!ok) { code.clear(); (void)unlink(file.c_str()); }
  return ok;
}

void cache_write(const std::string &file, const std::vector<char> &code) {
  const std::string tmp = file + ".tmp" + std::to_string((long)getpid());
  FILE *fh = fopen(tmp.c_str(), "wb");
  if (!fh) return;
  const uint64_t head[3] = {CACHE_MAGIC, (uint64_t)code.size(), fnv1a_bytes(code.data(), code.size())};
  const bool ok = fwrite(head, sizeof(uint64_t), 3, fh) == 3 && fwrite(code.data(), 1, code.size(), fh) == code.size();
  if (fclose(fh) == 0 && ok) (void)rename(tmp.c_str(), file.c_str());
  else (void)unlink(tmp.c_str());
}

// from_cache (optional): *from_cache <- the code came from the disk cache; fresh: do not read the cache (rebuild and rewrite)
bool compile_source(const std::string &src, std::vector<char> &code, std::string &log, bool *from_cache = nullptr, bool fresh = false) {
  Rtc &r = rtc();
  if (from_cache) *from_cache = false;
  if (!r.ok) { log = r.why; return false; }
  const std::string file = cache_file(src);
  if (!file.empty() && !fresh && cache_read(file, code)) {
    if (from_cache) *from_cache = true;
    return true;
  }
  Prog p = nullptr;
  if (r.Create(&p, src.c_str(), "lsqamd_tape.hip", 0, nullptr, nullptr) != 0) { log = "hiprtcCreateProgram failed"; return false; }
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  const int rc = r.Compile(p, 3, opts);
  size_t n = 0;
  if (r.LogSize(p, &n) == 0 && n > 1) {
    log.resize(n);
    (void)r.Log(p, &log[0]);
  }
  if (rc != 0) {
    (void)r.Destroy(&p);
    if (log.empty()) log = "hiprtcCompileProgram failed";
    return false;
  }
  n = 0;
  if (r.CodeSize(p, &n) != 0 || n == 0) { (void)r.Destroy(&p); log = "hiprtc produced no code"; return false; }
  code.resize(n);
  const int rg = r.Code(p, code.data());
  (void)r.Destroy(&p);
  if (rg != 0) { log = "hiprtcGetCode failed"; return false; }
  if (!file.empty()) cache_write(file, code);
  return true;
}

// build (or fetch) and load: a cached object that does not load -- stale for this runtime, damaged in a way the checksum
// cannot see -- is removed and the source compiled afresh, once
hipError_t build_module(const std::string &src, hipModule_t *mod, std::string &log) {
  std::vector<char> obj;
  bool cached = false;
  if (!compile_source(src, obj, log, &cached)) return hipErrorInvalidSource;
  hipError_t e = hipModuleLoadData(mod, obj.data());
  if (e != hipSuccess && cached) {
    (void)hipGetLastError();
    const std::string file = cache_file(src);
    if (!file.empty()) (void)unlink(file.c_str());
    if (!compile_source(src, obj, log, nullptr, true)) return hipErrorInvalidSource;
    e = hipModuleLoadData(mod, obj.data());
  }
  if (e != hipSuccess) { (void)hipGetLastError(); log = "the compiled tape could not be loaded"; }
  return e;
}

struct Loaded {
  hipModule_t mod = nullptr;
  hipFunction_t res = nullptr, jac = nullptr, nrm = nullptr, lm = nullptr, lmb = nullptr;
  bool wave_per_row = false;
  int n_param = 0;
  bool nrm_ok = false;          // few parameters, no wide sums: a third kernel forms J^T J, J^T f and chi2 without writing J
};
std::mutex g_mu;

}  // namespace

namespace lsqamd_jit {

struct Kernel {
  Loaded l;
  // what it takes to build the batched whole-fit kernel later (ensure_batch): a module of its own, compiled the first time a
  // batch of fits asks for it -- inlined into the main module it doubled every formula's build time
  std::vector<int32_t> code;
  std::vector<double> consts;
  int n_x = 1;
  bool fit_ok = false, batch_tried = false;
  hipModule_t modb = nullptr;
  int refs = 0;               // holders (guarded by g_mu)
  unsigned long long used = 0;   // cache clock at the last compile_tape that returned it
};

int plan_and_generate(const int32_t *code, int n_code, const double *consts, int n_consts, int P, int n_x,
                      std::string &src, int *variant, std::string &why, bool *has_nrm = nullptr, bool *has_fit = nullptr) {
  Plan pl;
  if (!make_plan(code, n_code, consts, n_consts, P, n_x, pl, why)) return 1;
  src = generate(pl);
  if (variant) *variant = pl.wave_per_row ? 1 : 0;
  if (has_nrm) *has_nrm = pl.nrm_ok;
  if (has_fit) *has_fit = pl.fit_ok;
  return 0;
}

namespace {
typedef std::tuple<int, uint64_t, uint64_t, size_t> SrcKey;
std::map<std::string, Kernel *> g_by_tape;          // tape bytes -> kernel (shortcut past planning and code generation)
std::map<SrcKey, Kernel> g_kernels;                 // generated source (two hashes + length) -> loaded kernel
unsigned long long g_clock = 0, g_evicted = 0;

size_t cache_cap() {
  const char *e = getenv("LSQAMD_JIT_CACHE_CAP");
  const long v = e ? atol(e) : 0;
  return v > 0 ? (size_t)v : (size_t)1024;
}

// (g_mu held) unload kernels nobody holds, least recently used first, until the cache is three quarters full
void evict_unheld() {
  const size_t cap = cache_cap();
  if (g_kernels.size() < cap) return;
  std::vector<std::pair<unsigned long long, SrcKey>> idle;
  for (auto &kv : g_kernels)
    if (kv.second.refs == 0) idle.push_back({kv.second.used, kv.first});
  std::sort(idle.begin(), idle.end());
  size_t target = cap - cap / 4;
  for (size_t i = 0; i < idle.size() && g_kernels.size() > target; ++i) {
    auto it = g_kernels.find(idle[i].second);
    Kernel *k = &it->second;
    for (auto bt = g_by_tape.begin(); bt != g_by_tape.end();) bt = bt->second == k ? g_by_tape.erase(bt) : std::next(bt);
    if (k->modb) (void)hipModuleUnload(k->modb);
    if (k->l.mod) (void)hipModuleUnload(k->l.mod);
    g_kernels.erase(it);
    ++g_evicted;
  }
}
}  // namespace

void release(const Kernel *kc) {
  if (!kc) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Kernel *k = const_cast<Kernel *>(kc);
  if (k->refs > 0) --k->refs;
  k->used = ++g_clock;      // (just released = recently used: not the first candidate of the next eviction)
}

void cache_stats(long long out[3]) {
  std::lock_guard<std::mutex> lk(g_mu);
  long long held = 0;
  for (auto &kv : g_kernels) held += kv.second.refs > 0;
  out[0] = (long long)g_kernels.size(); out[1] = held; out[2] = (long long)g_evicted;
}

const Kernel *compile_tape(const int32_t *code, int n_code, const double *consts, int n_consts, int P, int n_x, std::string &why) {
  if (!rtc().ok) { why = rtc().why; return nullptr; }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { why = "no device"; (void)hipGetLastError(); return nullptr; }
  // the same tape again (a fit per data set with one formula: the usual case): found by the tape's own bytes, before any
  // planning or code generation (~0.1 ms for a short formula -- as long as the fit itself when that takes one launch)
  std::string tkey(reinterpret_cast<const char *>(code), sizeof(int32_t) * (size_t)n_code);
  tkey.append(reinterpret_cast<const char *>(consts), sizeof(double) * (size_t)(n_consts > 0 ? n_consts : 0));
  const int dims[3] = {P, n_x, dev};
  tkey.append(reinterpret_cast<const char *>(dims), sizeof(dims));
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto bt = g_by_tape.find(tkey);
    if (bt != g_by_tape.end()) {
      ++bt->second->refs;
      bt->second->used = ++g_clock;
      return bt->second;
    }
  }
  std::string src;
  int variant = 0;
  bool has_nrm = false, has_fit = false;
  if (plan_and_generate(code, n_code, consts, n_consts, P, n_x, src, &variant, why, &has_nrm, &has_fit)) return nullptr;
  // (the source itself is not kept: two independent 64-bit hashes and the length stand for it)
  const SrcKey key{dev, fnv1a(src), fnv1a(src, 88172645463325252ull), src.size()};
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_kernels.find(key);
  if (it != g_kernels.end()) {
    if (g_by_tape.size() < 4096) g_by_tape[tkey] = &it->second;
    ++it->second.refs;
    it->second.used = ++g_clock;
    return &it->second;
  }
  evict_unheld();
  std::string log;
  Kernel k;
  const hipError_t be = build_module(src, &k.l.mod, log);
  if (be == hipErrorInvalidSource) { why = "hiprtc: " + log.substr(0, 400); return nullptr; }
  if (be != hipSuccess ||
      hipModuleGetFunction(&k.l.res, k.l.mod, "lsqamd_jit_res") != hipSuccess ||
      hipModuleGetFunction(&k.l.jac, k.l.mod, "lsqamd_jit_jac") != hipSuccess) {
    (void)hipGetLastError();
    if (k.l.mod) (void)hipModuleUnload(k.l.mod);
    why = "the compiled tape could not be loaded";
    return nullptr;
  }
  k.l.wave_per_row = variant == 1;
  k.l.n_param = P;
  if (has_nrm && hipModuleGetFunction(&k.l.nrm, k.l.mod, "lsqamd_jit_nrm") != hipSuccess) {
    (void)hipGetLastError();
    k.l.nrm = nullptr;
  }
  if (has_fit && hipModuleGetFunction(&k.l.lm, k.l.mod, "lsqamd_jit_lm") != hipSuccess) {
    (void)hipGetLastError();
    k.l.lm = nullptr;
  }
  k.fit_ok = has_fit && k.l.lm != nullptr;
  k.code.assign(code, code + n_code);
  k.consts.assign(consts, consts + (n_consts > 0 ? n_consts : 0));
  k.n_x = n_x;
  k.refs = 1;
  k.used = ++g_clock;
  Kernel *kp = &g_kernels.emplace(key, k).first->second;
  if (g_by_tape.size() < 4096) g_by_tape[tkey] = kp;
  return kp;
}

hipError_t launch(const Kernel *k, hipStream_t st, bool jac, const LaunchArgs &a) {
  if (a.n_data <= 0) return hipSuccess;
  struct { const double *x, *p, *ymean, *wdiag; const unsigned char *in_block; double *out_w, *out_raw; long long ld, n_data;
           long long p_stride, out_stride, ymean_stride; const int *batch_active; } args =
      {a.x, a.p, a.ymean, a.wdiag, a.in_block, a.out_w, a.out_raw, (long long)a.ld, (long long)a.n_data,
       (long long)a.p_stride, (long long)a.out_stride, (long long)a.ymean_stride, a.batch_active};
  size_t sz = sizeof(args);
  void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  int64_t blocks = k->l.wave_per_row ? (a.n_data + 3) / 4 : (a.n_data + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  return hipModuleLaunchKernel(jac ? k->l.jac : k->l.res, (unsigned)blocks, (unsigned)(a.n_batch < 1 ? 1 : a.n_batch), 1, 256, 1, 1, 0,
                               st, nullptr, cfg);
}

int normal_nq(const Kernel *k) {
  if (!k || !k->l.nrm) return 0;
  const int P = k->l.n_param;
  return P * (P + 1) / 2 + P + 1;
}

hipError_t launch_normal(const Kernel *k, hipStream_t st, const LaunchArgs &a, double *partial, int blocks) {
  if (!k || !k->l.nrm || blocks < 1) return hipErrorInvalidValue;
  struct { const double *x, *p, *ymean, *wdiag; const unsigned char *in_block; double *out_w, *out_raw; long long ld, n_data;
           long long p_stride, out_stride, ymean_stride; const int *batch_active; } args =
      {a.x, a.p, a.ymean, a.wdiag, nullptr, partial, nullptr, 1, (long long)a.n_data, 0, 0, 0, nullptr};
  size_t sz = sizeof(args);
  void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(k->l.nrm, (unsigned)blocks, 1, 1, 256, 1, 1, 0, st, nullptr, cfg);
}

bool has_fit_kernel(const Kernel *k) { return k && k->l.lm; }
int64_t fit_row_limit(const Kernel *k, bool correlated) {
  if (!k || !k->l.lm) return 0;
  const int P = k->l.n_param;
  if (correlated) return P > NRM_MAX_P ? fit_wide_rows(P) : FIT_MAX_BLOCK_ROWS;
  if (P <= NRM_MAX_P) return FIT_MAX_ROWS;
  // wide form: one workgroup adds up N products for each of P (P + 1) / 2 + P + 1 sums -- beyond ~400 000 of them per evaluation
  // the general path's many workgroups are faster (P = 16, N = 1000: 0.43 ms here, 1.08 there; P = 32, N = 1000: 3.4 against 2.5)
  const int64_t cap = 400000 / (P * (P + 1) / 2 + P + 1);
  return cap < FIT_MAX_ROWS ? cap : FIT_MAX_ROWS;
}
bool has_batch_fit_kernel(const Kernel *kc) {
  if (!kc || !kc->fit_ok) return false;
  Kernel *k = const_cast<Kernel *>(kc);      // (the cache owns the object; its batch half is filled in under the lock)
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (k->batch_tried) return k->l.lmb != nullptr;     // (a build in flight on another thread: that call takes the lockstep engine)
    k->batch_tried = true;
  }
  // the build itself (0.9 - 1.6 s of hiprtc for a new formula) runs outside the cache's lock; the caller holds the kernel
  Plan pl;
  std::string why, log;
  hipModule_t modb = nullptr;
  hipFunction_t lmb = nullptr;
  if (!(make_plan(k->code.data(), (int)k->code.size(), k->consts.data(), (int)k->consts.size(), k->l.n_param, k->n_x, pl, why) &&
        pl.fit_ok && build_module(generate(pl, true), &modb, log) == hipSuccess &&
        hipModuleGetFunction(&lmb, modb, "lsqamd_jit_lmb") == hipSuccess)) {
    (void)hipGetLastError();
    if (modb) (void)hipModuleUnload(modb);
    modb = nullptr;
    lmb = nullptr;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  k->modb = modb;
  k->l.lmb = lmb;
  return lmb != nullptr;
}

hipError_t launch_fit_batch(const Kernel *k, hipStream_t st, const FitArgs &a, const FitBatch &b, int n_fits) {
  if (!k || !k->l.lmb || n_fits < 1) return hipErrorInvalidValue;
  struct { FitArgs a; FitBatch b; } args = {a, b};
  size_t sz = sizeof(args);
  void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(k->l.lmb, (unsigned)n_fits, 1, 1, 256, 1, 1, 0, st, nullptr, cfg);
}

hipError_t launch_fit(const Kernel *k, hipStream_t st, const FitArgs &a) {
  if (!k || !k->l.lm) return hipErrorInvalidValue;
  FitArgs args = a;
  size_t sz = sizeof(args);
  void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(k->l.lm, 1, 1, 1, 256, 1, 1, 0, st, nullptr, cfg);
}

bool available(std::string *why) {
  if (why) *why = rtc().ok ? rtc().where : rtc().why;
  return rtc().ok;
}

}  // namespace lsqamd_jit

extern "C" int lsqamd_jit_cache_stats(int64_t *out3) try {
  if (!out3) return LSQAMD_EINVAL;
  long long v[3];
  lsqamd_jit::cache_stats(v);
  for (int i = 0; i < 3; ++i) out3[i] = v[i];
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)

extern "C" int lsqamd_tape_codegen(const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts, int32_t n_param,
                                   int32_t n_x, char *src_out, size_t cap, int32_t *variant, int32_t compile) try {
  if (!code || n_code < 1 || n_param < 0) return LSQAMD_EINVAL;
  for (int t = 0; t < n_code; ++t) {
    const int op = code[t] & 0xff, arg = code[t] >> 8;
    if ((op == LSQAMD_OP_P && (arg < 0 || arg >= n_param)) || (op == LSQAMD_OP_CONST && (arg < 0 || arg >= n_consts)) ||
        (op == LSQAMD_OP_X && (arg < 0 || arg >= (n_x < 1 ? 1 : n_x))) || op > LSQAMD_OP_LAST)
      return LSQAMD_EINVAL;
  }
  std::string src, why;
  int v = 0;
  if (lsqamd_jit::plan_and_generate(code, n_code, consts, n_consts, n_param, n_x, src, &v, why)) {
    if (src_out && cap) snprintf(src_out, cap, "%s", why.c_str());
    return LSQAMD_EUNSUPPORTED;
  }
  if (variant) *variant = v;
  if (src_out && cap) snprintf(src_out, cap, "%s", src.c_str());
  if (compile) {
    std::vector<char> obj;
    std::string log;
    if (!compile_source(src, obj, log)) {
      if (src_out && cap) snprintf(src_out, cap, "%s", log.c_str());
      return rtc().ok ? LSQAMD_EHIP : LSQAMD_EUNSUPPORTED;
    }
  }
  return 0;
} LSQAMD_ABI_CATCH(return lsqamd::abi_exception(nullptr);)
