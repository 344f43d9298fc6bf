// Persistent decode "chain": the GEMMs of a run of HBM-resident decoder layers, their split-K combines and the row ops between
// them (bias / residual / LayerNorm / RMSNorm / RoPE / SiLU-gate / KV-cache scatter) as ONE launch per layer, one workgroup per CU.
// Host-side description of a chain program (lia_chain.hip runs it; lia_api.hip builds it).
#pragma once
#include "lia_common.h"

enum { LIA_CH_GEMM = 1, LIA_CH_REDUCE_NORM = 2, LIA_CH_REDUCE_MAP = 3 };
enum { LIA_CH_DIRECT_NONE = 0, LIA_CH_DIRECT_PLAIN = 1, LIA_CH_DIRECT_GLU = 2 };
#define LIA_CHAIN_MAX_OPS 8      /* the program travels as a kernel argument: 8 x 432 B + the rest + the hidden arguments stay under 4 KB */

// One step of the program.  Every step is followed by a grid barrier (all workgroups, agent-scope hand-off), so a step may
// read whatever an earlier step of the same launch wrote.
struct LiaChainOp {
  int kind;               // LIA_CH_*
  int M, N, K;            // GEMM: y[M,N] = x[M,K] . W[N,K]^T;  REDUCE_*: the [M,N] output being combined
  // ---- GEMM ----
  const bf16_t* x;        // [M][ldx]
  long ldx;
  const bf16_t* W;        // [N][ldw]
  long ldw;
  int bn;                 // weight rows per work item (multiple of 16)
  int split, cps;         // K slices and 64-column chunks per slice (the last slice may be shorter); nchunks = K / 64
  int nchunks;
  int n_items;            // ceil(N / bn) * split; item i = (tile i / split, slice i % split), workgroup b takes items b, b + G, ...
  int direct;             // LIA_CH_DIRECT_*: split == 1 only -- the finished tile leaves through ep / om instead of a slab
  // ---- slabs: written by a split GEMM, read by the REDUCE step behind it ----
  float* slab;            // [slices][M][N] fp32
  int slices;
  int use_pos0;           // REDUCE_MAP: om.pos0 / post.pos0 are replaced by the launch's pos0 argument (decode position)
  LiaEpilogue ep;         // bias / residual / relu (/ glu) of the finished values
  LiaOutMap om;           // where the finished bf16 values go (segments, KV-cache scatter)
  LiaPost post;           // REDUCE_NORM: LIA_POST_LAYERNORM / RMSNORM (or NONE) of the finished row into post.out; REDUCE_MAP: NONE / ROPE
};

struct LiaChainProgram {
  int n_ops;
  int pad_;
  LiaChainOp op[LIA_CHAIN_MAX_OPS];
};

struct LiaChainPlan { int bn, split, cps; };

extern "C" {
// work decomposition of one decode GEMM on `n_cu` workgroups: rows per item, K slices (cdna: one item per CU wherever N allows)
int lia_chain_plan_gemm(int M, int N, int K, int glu, int n_cu, LiaChainPlan* out);
// geometry the chain kernels support for M rows: returns 0 and the LDS bytes, or -1
int lia_chain_supported(int M);
// launch a program (host memory; copied into the kernel arguments); sync_block: 4 KB of zeroed device memory this
// launch owns exclusively (barrier counters + timeout word at word 17 * 32); err_host: host-mapped word that a barrier which
// gave up sets (nullable); gran: LIA_CHAIN_GRAN_BYTES of device memory for the row statistics that cross workgroups (nullable: one
// workgroup per row then), never zeroed -- epoch must differ from launch to launch (a counter)
int lia_chain_launch(const LiaChainProgram* prog, int M, unsigned* sync_block, unsigned* err_host, int pos0, int n_cu, unsigned long long* gran,
                     unsigned epoch, hipStream_t st);
int lia_chain_cu_count(int device);
}
#define LIA_CHAIN_SYNC_BYTES 4096
#define LIA_CHAIN_GRAN_BYTES (2 * 128 * 16 * 8)
#define LIA_CHAIN_ERR_WORD (17 * 32)
