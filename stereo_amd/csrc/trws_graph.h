// Host-side graph analysis for the TRW-S path (product code, C++).
//
// Produces, for an arbitrary directed edge list, exactly the node order, edge
// orientation and per-node forward/backward edge lists that the reference
// builds in MRFEnergy::AddEdge (cpp/trw-s/MRFEnergy.cpp:83-111),
// SetAutomaticOrdering (cpp/trw-s/ordering.cpp:7-157) and
// CompleteGraphConstruction (cpp/trw-s/MRFEnergy.cpp:137-229), plus the
// dependency levels of that order that the level-synchronous HIP sweeps run on.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace stereo {

struct TrwsGraph {
  int64_t N = 0, E = 0;
  // per edge, after orientation: tail has the lower rank
  std::vector<int32_t> tail, head;
  std::vector<uint8_t> mdir;  // Swap() parity (typeStereoLinear.h:318-321)
  // node order
  std::vector<int32_t> order;  // order[r] = node id
  std::vector<int32_t> rank;   // rank[node]
  // per RANK r: forward / backward edge ids in the reference's list order (CSR)
  std::vector<int32_t> fptr, fidx, bptr, bidx;
  std::vector<double> gamma;  // per rank: 1/max(nFwd,nBwd) (treeProbabilities.cpp:24-45)
  // dependency levels of the forward sweep (longest path over backward edges)
  std::vector<int32_t> level_ptr;    // size L+1
  std::vector<int32_t> level_ranks;  // ranks grouped by level, ascending rank inside
  int64_t max_level_nodes = 0;
  // position of every lower-bound term in the reference's summation order
  // (minimize.cpp:67-95: node minimum, then one term per backward edge, nodes
  // visited in descending rank); energy terms are simply indexed by rank.
  std::vector<int32_t> lb_pos_node;  // per rank
  std::vector<int32_t> lb_pos_edge;  // per edge
  int64_t lb_terms = 0;

  // Row strips (multi-GPU): owner[node] = strip that visits the node; empty = one strip.  Strips
  // form a chain: every edge joins nodes of the same strip or of strips g, g+1.  With strips the
  // positions above are STRIP-LOCAL (every strip sums its own terms, in the reference's order
  // restricted to its nodes; the partial sums are combined in strip order), runs never cross a
  // strip boundary, and the descriptors say which outgoing messages / flags / labels go to the
  // neighbouring strip's memory.
  std::vector<int32_t> owner;            // per NODE
  int nstrips = 1;
  std::vector<int64_t> strip_lb_terms;   // per strip
  std::vector<int64_t> strip_nodes;      // per strip
  std::vector<int32_t> e_pos;            // per rank: position of the node's energy term in its strip

  // Schedule of the persistent dataflow sweeps.  Processing position p is rank p
  // in the forward sweep and rank N-1-p in the backward sweep.  A "run" is a
  // maximal stretch of consecutive positions in which every node has its
  // predecessor among its incoming neighbours (a grid row, the border chain):
  // one workgroup walks a run sequentially, hands the messages for the next
  // node over in LDS and waits on completion flags only for `dep_rank`.
  struct Sweep {
    std::vector<int32_t> run_ptr;   // R+1 offsets into processing positions
    // order in which workgroups pick runs up (ticket -> run); empty = 0, 1, 2, ...
    std::vector<int32_t> run_order;
    std::vector<int32_t> dep_ptr;   // N+1, indexed by rank
    std::vector<int32_t> dep_rank;  // ranks whose flag must be set first
    // aligned with the node's INCOMING list (bidx forward / fidx backward):
    // slot of that edge in the predecessor's outgoing list, or -1
    std::vector<int8_t> in_slot;
    // Packed per-position descriptors for the fast kernels (kDescWords int32 each,
    // layout in trws.hip: NodeDesc); empty unless fast_ok.  They are laid out in the order of
    // the CHAIN schedule below, which is what the descriptor-driven kernels walk.
    std::vector<int32_t> desc;
    // Chain schedule of the fast kernels: a run is a path of the dependency DAG (every node
    // hangs on the node visited just before it), not necessarily a stretch of consecutive
    // ranks -- the reference order interleaves the ranks of the last two grid rows, which a
    // rank-contiguous run would walk as ONE serial chain of 2W visits.  chain_rank maps a
    // schedule position to the rank visited there; chain_run_ptr / chain_run_order are the
    // counterparts of run_ptr / run_order over schedule positions.
    std::vector<int32_t> chain_rank, chain_run_ptr, chain_run_order;
    // strip that owns each run of the rank-contiguous / chain schedule (empty with one strip)
    std::vector<int32_t> run_strip, chain_run_strip;
    // Speculative schedule.  The image grid's border chain is ONE run of 2(H+W)-4 strictly
    // serial visits during most of which nothing else can run.  What makes it serial is a single row per
    // visit -- the message a node hands to the next one -- and that row is plain min-plus of the previous
    // one whenever the certificate holds.  So the run is cut into segments of `seg_len` visits; a RUNNER
    // (one wave of one workgroup) walks the whole run computing nothing but that recurrence, uncertified,
    // and leaves the row at every cut; the segments -- ordinary runs, every visit exact and certified --
    // start from the runner's rows side by side on as many workgroups, hold their completion flags back,
    // and COMMIT in order: a segment compares the rows it started from with what the segment in front
    // really produced (bit for bit) and walks its visits a second time if they differ.  By induction every
    // committed value is the sequential sweep's.  The schedule below is the chain schedule with that one
    // run replaced by a runner ticket + its segments; descriptors are shared (the kernel treats the first
    // visit of a segment differently, nothing in the descriptor says so).
    struct Spec {
      bool ok = false;
      int32_t run = -1;                  // the run of the chain schedule that is cut
      int32_t c0 = 0, c1 = 0;            // its schedule positions
      int32_t seg_len = 0, nseg = 0, max_len = 0;
      std::vector<int32_t> run_ptr;      // CSR over schedule positions, the cut run as nseg runs
      std::vector<int32_t> run_order;    // ticket -> run; -1: the runner's ticket (ticket 0)
      std::vector<int32_t> kind;         // per run: 0, or 1 + segment index
    } spec;
  } sweep[2];
  static constexpr int kDescWords = 64;
  // every node has <= 8 incident edges and <= 4 foreign dependencies per direction
  bool fast_ok = false;
  static constexpr int kMaxSlots = 8;
};

// conn: 2 x E zero-based (column major: conn[2e] = tail, conn[2e+1] = head).
// Returns false and sets `err` on invalid input.
// max_resident_runs > 0: if a sweep has more runs than that, cut runs in front of nodes that
// wait long for a foreign node and dispense them in dependency-level order (see trws_graph.cpp).
// owner (N entries, values 0 .. nstrips-1) or nullptr: see TrwsGraph::owner.
// certainly_resident: workgroups sure to be resident whatever the kernel's LDS use (the CU count).
bool build_trws_graph(int64_t N, int64_t E, const uint32_t *conn, TrwsGraph &g,
                      std::string &err, int64_t max_resident_runs = 0,
                      const int32_t *owner = nullptr, int nstrips = 1, int64_t certainly_resident = 256,
                      int ordering = 0);
// visits per segment of the speculative schedule (STEREO_HIP_TRWS_SPEC_SEG, default 16)
int spec_segment_length();
// ordering: 0 = SetAutomaticOrdering (ordering.cpp:7-157, what the gateway calls, trws_mex.cpp:121);
// 1 = node index order, MRFEnergy's order when SetAutomaticOrdering is not called (nodes in the
// order they were added, MRFEnergy.cpp:37-76).  On the image grid: H + W - 1 anti-diagonal levels,
// no serial border chain.  Another valid TRW-S schedule, not the gateway's results.

// Descriptor words added for strips (layout of the rest: trws.hip NodeDesc)
//   word 43: bits 0-7 outgoing message k goes to the neighbouring strip; bits 8-15 which one
//            (0 = previous strip, 1 = next strip); bit 16 / 17: flag (+ label) also raised in the
//            previous / next strip's memory
//   word 44: TrwsGraph::e_pos of the node
constexpr int kDescRemote = 43, kDescEpos = 44;
//   words 45-52: with strip-local storage, the id of outgoing edge k in the neighbouring strip's
//            numbering (valid where bit k of word 43 is set)
//   words 53, 54: the node's id in the previous / next strip's numbering (bits 16 / 17 of word 43)
constexpr int kDescPeerEdge = 45, kDescPeerNode = 53;
//   word 55: bits 0-7: edge k of the node's list (k >= n_out: an incoming edge) brings its message from GLOBAL memory --
//            the other end was not visited one or two steps earlier in the same run, so a loader fetches the row (and
//            that node's label) behind the node's completion flag
constexpr int kDescFetch = 55;
//   word 56: nibble k: the outgoing message that goes to the same neighbour as outgoing message k (k itself: none)
constexpr int kDescTwin = 56;

// Strip-local storage.  A strip keeps arrays only for what it touches: its own nodes, the nodes
// one edge away (whose flags it waits on and whose labels its primal pass reads), and the edges
// with an own endpoint.  Local ids: own nodes in ascending global id, then the halo nodes in
// ascending global id; edges in ascending global id.  A flag is indexed by the local NODE id.
// The descriptors of the strip's own visits are renumbered accordingly and laid out run after
// run in ticket order, so the kernels see an ordinary single-strip problem of the local size
// plus the peer ids of words 45-54.
struct StripLayout {
  int64_t n_own = 0;
  std::vector<int32_t> nodes, edges;  // local id -> global id
  std::vector<int32_t> desc[2], run_ptr[2];
  bool need_peer[2] = {false, false};
};
bool build_strip_layout(const TrwsGraph &g, int strip, StripLayout &out, std::string &err);

}  // namespace stereo
