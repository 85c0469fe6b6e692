// multigpu.h -- the first pass with the Bloom filter sharded by bit address over several GPUs of one node
// (BASELINE.json north_star; the shared filter it replaces: reference graphconstructor/concurrentbitvector.cpp:31-52,
// filled by FilterFillerWorker vertexenumerator.h:995-1105 and read by CandidateCheckingWorker :586-704).
//
// One host thread per GPU ("rank"), each with its own context of the device library; the library never communicates
// (include/twopaco_hip.h, tpc_shard_*), this layer moves the device buffers between the ranks:
//   RcclTransport      ncclSend / ncclRecv groups and ncclAllGather over xGMI (one communicator per device,
//                      ncclCommInitAll); librccl is loaded on first use, so single-GPU runs never pay for it
//   LoopbackTransport  device-to-device copies between the ranks of this process: the fallback transport, and the way the
//                      whole orchestration is tested with several ranks on ONE device (RCCL refuses duplicate devices)
// Host-side scalars (counts, overflow sizes) travel through the process's own memory.
#ifndef _TPC_MULTIGPU_H_
#define _TPC_MULTIGPU_H_

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <chrono>
#include <string>
#include <utility>

struct tpc_ctx;

namespace TwoPaCo
{
	// Reusable barrier for the rank threads; also carries the first error so that no rank waits for a dead one.
	class RankBarrier
	{
	public:
		explicit RankBarrier(int ranks) : ranks_(ranks), waiting_(0), generation_(0), failed_(false) {}
		void Wait();                         // throws if any rank reported a failure
		void Fail(const std::string & what);  // wakes everyone
		bool Failed() const { return failed_; }
		const std::string & Error() const { return error_; }
	private:
		int ranks_, waiting_;
		uint64_t generation_;
		bool failed_;
		std::string error_;
		std::mutex mutex_;
		std::condition_variable cv_;
	};

	// Collectives on DEVICE buffers; every rank calls the same sequence, a call returns when the rank's data is in place.
	class Transport
	{
	public:
		Transport(int ranks) : ranks_(ranks), barrier_(ranks), scratch_(size_t(ranks) * 64) {}
		virtual ~Transport() {}
		int Ranks() const { return ranks_; }
		// block d of `send` goes to rank d; block s of `recv` came from rank s.  skipSelf: the caller reads its own block from `send`
		// (tpc_shard_apply_inplace), block `rank` of `recv` is left alone
		virtual void AllToAll(int rank, const void * send, void * recv, size_t blockBytes, bool skipSelf = false) = 0;
		// sendCounts[d] elements for rank d (contiguous, rank order); recvCounts[s] elements from rank s
		virtual void AllToAllV(int rank, const void * send, const uint64_t * sendCounts, void * recv, const uint64_t * recvCounts, size_t elemBytes) = 0;
		virtual void AllGather(int rank, const void * send, void * recv, size_t bytes) = 0;
		virtual const char * Name() const = 0;
		// n host values per rank -> all[r * n + i] on every rank (n <= 64)
		void ExchangeHost(int rank, const uint64_t * mine, int n, std::vector<uint64_t> & all);
		// any number of host values per rank (the same number on every rank) -> all[r] = rank r's values, on every rank
		void GatherHost(int rank, const std::vector<uint64_t> & mine, std::vector<std::vector<uint64_t> > & all);
		RankBarrier & Barrier() { return barrier_; }
		uint64_t BytesMoved() const { return bytesMoved_; }
	protected:
		int ranks_;
		RankBarrier barrier_;
		std::vector<uint64_t> scratch_;
		std::vector<std::vector<uint64_t> > gather_;
		uint64_t bytesMoved_ = 0;  // by rank 0
	};

	// devices[r] = HIP device of rank r.  rccl = false, or a device listed twice: LoopbackTransport.
	std::unique_ptr<Transport> MakeTransport(const std::vector<int> & devices, bool rccl);

	// One round of the address-sharded first pass on rank `rank` (all ranks call it with the same lo/hi): insert, query,
	// verification of the survivors, union of the candidate masks.  Afterwards every rank's context holds the round mask
	// tpc_pass1_query would have produced on one GPU.  marks = set bits of that mask.
	struct ShardedRank
	{
		int rank;
		int device;
		tpc_ctx * ctx;
		// device scratch owned by the rank (grown on demand, freed by Release)
		enum { BUFFERS = 24 };
		void * buf[BUFFERS];
		size_t cap[BUFFERS];
		// exact-size exchange of the level-1 regions (tpc_shard_pack / tpc_shard_apply_packed); false: equal blocks
		bool compactExchange;
		// second pass with the exact filter's table sharded by key hash (ShardedSecondPass; the text is then sharded on every
		// rank); false: union of the candidate masks, the single-GPU second pass on rank 0, which keeps the whole text
		bool shardedSecondPass;
		uint64_t regionBytesSent;
		bool filterLoaded;  // this round's filter shard was restored from a checkpoint (tpc_filter_upload): ShardedFirstPass skips the insert
		int verifyEager;  // survivor verification: -1 not decided yet, 0 lazy (function 1 alone, then the rest), 1 all q - 1 functions in one round trip
		// The combined exchange (include/twopaco_hip.h: tpc_combine_*): the context keeps the WHOLE filter (option replicate_filter), the rank
		// inserts its chunk locally, only the set bits of every slice travel (2 bytes per distinct bit), the query is local.  false: the
		// level-1 entries of both passes are routed to the owners of their slices (tpc_shard_*: filters no single GPU holds)
		bool combined;
		int filterBits;                 // L (the bytes model of the combined exchange)
		std::string combineMode;        // what the last combined round did: "gather", "scatter" or "dense"
		uint64_t combineBytesReceived;  // ... and the bytes this rank received for it
		// TWOPACO_TIMING=1: host-clock milliseconds per phase of the first pass (every phase ends synchronised), printed by rank 0
		std::vector<std::pair<std::string, double> > phaseMs;
		std::chrono::steady_clock::time_point phaseT0;
		bool phaseOn;
		void PhaseBegin();
		void Phase(const char * name);  // time since the previous Phase / PhaseBegin goes to `name`
		void PhasePrint(const char * title);
		ShardedRank() : rank(0), device(0), ctx(0), compactExchange(true), shardedSecondPass(true), regionBytesSent(0), filterLoaded(false), verifyEager(-1), combined(false), filterBits(0), combineBytesReceived(0), phaseOn(false) { for (int i = 0; i < BUFFERS; i++) { buf[i] = 0; cap[i] = 0; } }
		void * Ensure(int which, size_t bytes);
		void Release();
	};

	void ShardedFirstPass(ShardedRank & r, Transport & net, int hashFunctions, uint64_t lo, uint64_t hi);

	// The round's second pass when r.shardedSecondPass (CandidateFinalFilteringWorker + TrueBifurcations, reference
	// vertexenumerator.h:708-829, 1228-1256, with the key table cut over the ranks by key hash): every rank turns its own marks
	// into (canonical key, prev | next) records, the records travel to the key owners, each owner runs the exact filter over all
	// occurrences of its keys.  counters = {true junctions, false junctions, table size, marks} of THIS rank: the round's
	// figures are their sums over the ranks.
	void ShardedSecondPass(ShardedRank & r, Transport & net, uint64_t abundance, uint64_t counters[4]);

	// After the last round: every rank learns all junction keys (all-gather), sorts them (tpc_junctions_finalize: the same ids
	// everywhere), looks up the ids of its own marked positions, and the (position, id) lists are gathered on rank 0 in rank
	// order = position order, where they replace rank 0's list (tpc_emit_import) for the output stream.
	// gatherOnRankZero = false: the lists stay where they are (ShardedStream formats them rank by rank).
	void ShardedFinish(ShardedRank & r, Transport & net, uint64_t * junctions, bool gatherOnRankZero = true);

	// The junction stream formatted by every rank for its own chunk of the text (EdgeConstructionWorker's ordered flush,
	// reference vertexenumerator.h:837-854, and JunctionPositionWriter, junctionapi.h:118-126, as an exclusive scan over
	// per-sequence counts; include/twopaco_hip.h: tpc_emit_stream_partial / _part).  Afterwards the rank's context holds
	// bytes [firstByte, firstByte + nBytes) of the output file (tpc_emit_stream_fetch); records = junction occurrences +
	// stubs of the whole run (the reference's "True marks count").
	void ShardedStream(ShardedRank & r, Transport & net, const std::vector<uint64_t> & recStart, const std::vector<uint64_t> & recLength, size_t k,
		uint64_t * firstByte, uint64_t * nBytes, uint64_t * records);
}

#endif
