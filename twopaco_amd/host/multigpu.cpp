// multigpu.cpp -- see multigpu.h.  Orchestration of the address-sharded first pass (one thread per GPU) and its two
// transports.  The call sequence per pass is the one documented in include/twopaco_hip.h (tpc_shard_*):
//   hash (level 1 of the write-combining partition over this rank's tiles, regions grouped by destination rank)
//   -> all-to-all of the level-1 regions and their fill counts (ncclSend / ncclRecv group)
//   -> apply (levels 2-3 on the owned filter slices: OR for the insert, first Bloom probe for the query)
// and for the query: survivors of the first probe -> addresses of hash functions 1..q-1 routed to their owners
// (tpc_shard_route / _permute64, variable all-to-all), one answer byte back, tpc_shard_select, tpc_shard_mark; finally
// the candidate masks are OR-reduced by word ranges (all-to-all, fold, all-gather).
#include "multigpu.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <stdexcept>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../include/twopaco_hip.h"

namespace TwoPaCo
{
	namespace
	{
		void HipCheck(hipError_t e, const char * what)
		{
			if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
		}

		void LibCheck(tpc_ctx * ctx, int rc, const char * what)
		{
			if (rc != 0) throw std::runtime_error(std::string(what) + ": " + tpc_last_error(ctx));
		}
	}

	// ------------------------------------------------------------------------------------------ barrier + host exchange
	void RankBarrier::Wait()
	{
		std::unique_lock<std::mutex> lock(mutex_);
		if (failed_) throw std::runtime_error("another rank failed: " + error_);
		const uint64_t gen = generation_;
		if (++waiting_ == ranks_)
		{
			waiting_ = 0;
			++generation_;
			cv_.notify_all();
		}
		else
		{
			cv_.wait(lock, [&]() { return generation_ != gen || failed_; });
		}

		if (failed_) throw std::runtime_error("another rank failed: " + error_);
	}

	void RankBarrier::Fail(const std::string & what)
	{
		std::unique_lock<std::mutex> lock(mutex_);
		if (!failed_)
		{
			failed_ = true;
			error_ = what;
		}

		cv_.notify_all();
	}

	void Transport::ExchangeHost(int rank, const uint64_t * mine, int n, std::vector<uint64_t> & all)
	{
		if (n > 64) throw std::runtime_error("ExchangeHost: too many values");
		barrier_.Wait();  // the previous exchange has been read by everyone
		for (int i = 0; i < n; i++) scratch_[size_t(rank) * 64 + i] = mine[i];
		barrier_.Wait();
		all.resize(size_t(ranks_) * n);
		for (int r = 0; r < ranks_; r++)
		{
			for (int i = 0; i < n; i++) all[size_t(r) * n + i] = scratch_[size_t(r) * 64 + i];
		}
	}

	void Transport::GatherHost(int rank, const std::vector<uint64_t> & mine, std::vector<std::vector<uint64_t> > & all)
	{
		barrier_.Wait();  // the previous gather has been read by everyone
		if (rank == 0) gather_.resize(size_t(ranks_));
		barrier_.Wait();
		gather_[size_t(rank)] = mine;
		barrier_.Wait();
		all = gather_;
	}

	// ------------------------------------------------------------------------------------------ loopback transport
	namespace
	{
		class LoopbackTransport : public Transport
		{
		public:
			LoopbackTransport(const std::vector<int> & devices) : Transport(int(devices.size())), devices_(devices), send_(devices.size()), counts_(devices.size()) {}
			const char * Name() const { return "loopback (device-to-device copies inside the process)"; }

			void AllToAll(int rank, const void * send, void * recv, size_t blockBytes, bool skipSelf)
			{
				send_[rank] = send;
				barrier_.Wait();
				HipCheck(hipSetDevice(devices_[rank]), "hipSetDevice");
				for (int s = 0; s < ranks_; s++)
				{
					if (skipSelf && s == rank) continue;
					HipCheck(hipMemcpy(static_cast<char*>(recv) + size_t(s) * blockBytes, static_cast<const char*>(send_[s]) + size_t(rank) * blockBytes,
						blockBytes, hipMemcpyDeviceToDevice), "loopback all-to-all copy");
				}

				if (rank == 0) bytesMoved_ += uint64_t(ranks_ - (skipSelf ? 1 : 0)) * blockBytes;
				barrier_.Wait();  // nobody reuses a send buffer before every peer has copied from it
			}

			void AllToAllV(int rank, const void * send, const uint64_t * sendCounts, void * recv, const uint64_t * recvCounts, size_t elemBytes)
			{
				send_[rank] = send;
				counts_[rank].assign(sendCounts, sendCounts + ranks_);
				barrier_.Wait();
				HipCheck(hipSetDevice(devices_[rank]), "hipSetDevice");
				uint64_t at = 0;
				for (int s = 0; s < ranks_; s++)
				{
					uint64_t off = 0;
					for (int d = 0; d < rank; d++) off += counts_[s][d];
					if (counts_[s][rank] != recvCounts[s]) throw std::runtime_error("loopback all-to-all: count mismatch");
					if (recvCounts[s])
					{
						HipCheck(hipMemcpy(static_cast<char*>(recv) + at * elemBytes, static_cast<const char*>(send_[s]) + off * elemBytes,
							size_t(recvCounts[s]) * elemBytes, hipMemcpyDeviceToDevice), "loopback all-to-all copy");
					}

					at += recvCounts[s];
				}

				barrier_.Wait();
			}

			void AllGather(int rank, const void * send, void * recv, size_t bytes)
			{
				send_[rank] = send;
				barrier_.Wait();
				HipCheck(hipSetDevice(devices_[rank]), "hipSetDevice");
				for (int s = 0; s < ranks_; s++)
				{
					HipCheck(hipMemcpy(static_cast<char*>(recv) + size_t(s) * bytes, send_[s], bytes, hipMemcpyDeviceToDevice), "loopback all-gather copy");
				}

				barrier_.Wait();
			}

		private:
			std::vector<int> devices_;
			std::vector<const void*> send_;
			std::vector<std::vector<uint64_t> > counts_;
		};

		// -------------------------------------------------------------------------------------- RCCL transport
		struct RcclApi
		{
			void * lib;
			ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
			ncclResult_t (*CommDestroy)(ncclComm_t);
			ncclResult_t (*GroupStart)();
			ncclResult_t (*GroupEnd)();
			ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
			ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
			ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
			const char * (*GetErrorString)(ncclResult_t);
			RcclApi() : lib(0) {}
			bool Load()
			{
				lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
				if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
				if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
				if (!lib) return false;
				CommInitAll = reinterpret_cast<ncclResult_t (*)(ncclComm_t *, int, const int *)>(dlsym(lib, "ncclCommInitAll"));
				CommDestroy = reinterpret_cast<ncclResult_t (*)(ncclComm_t)>(dlsym(lib, "ncclCommDestroy"));
				GroupStart = reinterpret_cast<ncclResult_t (*)()>(dlsym(lib, "ncclGroupStart"));
				GroupEnd = reinterpret_cast<ncclResult_t (*)()>(dlsym(lib, "ncclGroupEnd"));
				Send = reinterpret_cast<ncclResult_t (*)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>(dlsym(lib, "ncclSend"));
				Recv = reinterpret_cast<ncclResult_t (*)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>(dlsym(lib, "ncclRecv"));
				AllGather = reinterpret_cast<ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>(dlsym(lib, "ncclAllGather"));
				GetErrorString = reinterpret_cast<const char * (*)(ncclResult_t)>(dlsym(lib, "ncclGetErrorString"));
				return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && AllGather && GetErrorString;
			}
		};

		class RcclTransport : public Transport
		{
		public:
			RcclTransport(const std::vector<int> & devices) : Transport(int(devices.size())), devices_(devices), comms_(devices.size()), streams_(devices.size())
			{
				if (!api_.Load()) throw std::runtime_error("Can't load librccl.so");
				Check(api_.CommInitAll(comms_.data(), ranks_, devices_.data()), "ncclCommInitAll");
				for (int r = 0; r < ranks_; r++)
				{
					HipCheck(hipSetDevice(devices_[r]), "hipSetDevice");
					HipCheck(hipStreamCreate(&streams_[r]), "hipStreamCreate");
				}
			}

			~RcclTransport()
			{
				for (int r = 0; r < ranks_; r++)
				{
					(void)hipSetDevice(devices_[r]);
					if (streams_[r]) (void)hipStreamDestroy(streams_[r]);
					if (comms_[r]) api_.CommDestroy(comms_[r]);
				}
			}

			const char * Name() const { return "RCCL (ncclSend/ncclRecv groups over xGMI)"; }

			// One ncclGroupStart ... ncclSend / ncclRecv per peer ... ncclGroupEnd moves at most CHUNK bytes per peer: multi-GiB
			// messages are cut up (a multi-GiB all_to_all arrived truncated in the torch.distributed driver; dist.py does the same).
			static const size_t CHUNK = size_t(1) << 28;

			void AllToAll(int rank, const void * send, void * recv, size_t blockBytes, bool skipSelf)
			{
				Agree();
				HipCheck(hipSetDevice(devices_[rank]), "hipSetDevice");
				for (size_t c0 = 0; c0 < blockBytes; c0 += CHUNK)
				{
					const size_t n = std::min(CHUNK, blockBytes - c0);
					Check(api_.GroupStart(), "ncclGroupStart");
					for (int p = 0; p < ranks_; p++)
					{
						if (skipSelf && p == rank) continue;
						Check(api_.Send(static_cast<const char*>(send) + size_t(p) * blockBytes + c0, n, ncclUint8, p, comms_[rank], streams_[rank]), "ncclSend");
						Check(api_.Recv(static_cast<char*>(recv) + size_t(p) * blockBytes + c0, n, ncclUint8, p, comms_[rank], streams_[rank]), "ncclRecv");
					}

					Check(api_.GroupEnd(), "ncclGroupEnd");
				}

				HipCheck(hipStreamSynchronize(streams_[rank]), "all-to-all");
				if (rank == 0) bytesMoved_ += uint64_t(ranks_ - (skipSelf ? 1 : 0)) * blockBytes;
			}

			void AllToAllV(int rank, const void * send, const uint64_t * sendCounts, void * recv, const uint64_t * recvCounts, size_t elemBytes)
			{
				Agree();
				HipCheck(hipSetDevice(devices_[rank]), "hipSetDevice");
				uint64_t most = 0, total = 0;
				for (int p = 0; p < ranks_; p++) { most = std::max(most, std::max(sendCounts[p], recvCounts[p])); total += sendCounts[p]; }
				const uint64_t step = std::max<uint64_t>(1, CHUNK / elemBytes);
				for (uint64_t c0 = 0; c0 < most; c0 += step)
				{
					Check(api_.GroupStart(), "ncclGroupStart");
					uint64_t so = 0, ro = 0;
					for (int p = 0; p < ranks_; p++)
					{
						const uint64_t sa = std::min(c0, sendCounts[p]), sb = std::min(c0 + step, sendCounts[p]);
						const uint64_t ra = std::min(c0, recvCounts[p]), rb = std::min(c0 + step, recvCounts[p]);
						if (sb > sa) Check(api_.Send(static_cast<const char*>(send) + (so + sa) * elemBytes, size_t(sb - sa) * elemBytes, ncclUint8, p, comms_[rank], streams_[rank]), "ncclSend");
						if (rb > ra) Check(api_.Recv(static_cast<char*>(recv) + (ro + ra) * elemBytes, size_t(rb - ra) * elemBytes, ncclUint8, p, comms_[rank], streams_[rank]), "ncclRecv");
						so += sendCounts[p];
						ro += recvCounts[p];
					}

					Check(api_.GroupEnd(), "ncclGroupEnd");
				}

				HipCheck(hipStreamSynchronize(streams_[rank]), "all-to-all");
				if (rank == 0) bytesMoved_ += total * elemBytes;
			}

			void AllGather(int rank, const void * send, void * recv, size_t bytes)
			{
				Agree();
				HipCheck(hipSetDevice(devices_[rank]), "hipSetDevice");
				Check(api_.AllGather(send, recv, bytes, ncclUint8, comms_[rank], streams_[rank]), "ncclAllGather");
				HipCheck(hipStreamSynchronize(streams_[rank]), "all-gather");
			}

		private:
			// Every rank agrees that nobody has failed before it enters a collective: a rank that threw between two collectives
			// (a library call returning an error on one rank only: a full survivor list, hipMalloc) never arrives here, its
			// Fail() wakes the ones that wait, and they leave with an exception instead of blocking forever in ncclGroupEnd.
			void Agree() { barrier_.Wait(); }

			void Check(ncclResult_t r, const char * what)
			{
				if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + api_.GetErrorString(r));
			}

			RcclApi api_;
			std::vector<int> devices_;
			std::vector<ncclComm_t> comms_;
			std::vector<hipStream_t> streams_;
		};
	}

	std::unique_ptr<Transport> MakeTransport(const std::vector<int> & devices, bool rccl)
	{
		std::vector<int> sorted(devices);
		std::sort(sorted.begin(), sorted.end());
		const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
		if (rccl && distinct) return std::unique_ptr<Transport>(new RcclTransport(devices));
		return std::unique_ptr<Transport>(new LoopbackTransport(devices));
	}

	// ------------------------------------------------------------------------------------------ one rank's scratch
	void * ShardedRank::Ensure(int which, size_t bytes)
	{
		if (bytes > cap[which])
		{
			HipCheck(hipSetDevice(device), "hipSetDevice");
			if (buf[which]) (void)hipFree(buf[which]);
			buf[which] = 0;
			cap[which] = 0;
			const size_t want = std::max<size_t>(bytes + bytes / 8, 256);
			HipCheck(hipMalloc(&buf[which], want), "hipMalloc (exchange buffer)");
			cap[which] = want;
		}

		return buf[which];
	}

	void ShardedRank::PhaseBegin()
	{
		phaseOn = rank == 0 && std::getenv("TWOPACO_TIMING") != 0;
		phaseT0 = std::chrono::steady_clock::now();
	}

	void ShardedRank::Phase(const char * name)
	{
		if (!phaseOn) return;
		const std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now();
		const double ms = std::chrono::duration<double, std::milli>(now - phaseT0).count();
		phaseT0 = now;
		for (auto & p : phaseMs) if (p.first == name) { p.second += ms; return; }
		phaseMs.push_back(std::make_pair(std::string(name), ms));
	}

	void ShardedRank::PhasePrint(const char * title)
	{
		if (!phaseOn) return;
		std::fprintf(stderr, "[timing]     %s (rank 0, ms):", title);
		for (auto & p : phaseMs) std::fprintf(stderr, " %s %.3f;", p.first.c_str(), p.second);
		size_t held = 0;
		for (int i = 0; i < ShardedRank::BUFFERS; i++) held += cap[i];
		std::fprintf(stderr, " region bytes sent %llu; exchange buffers held %.2f GB\n", (unsigned long long)regionBytesSent, double(held) / 1e9);
		regionBytesSent = 0;  // (per printed pass: the bench divides them by the pass' all-to-all time)
		phaseMs.clear();
	}

	void ShardedRank::Release()
	{
		(void)hipSetDevice(device);
		for (int i = 0; i < BUFFERS; i++)
		{
			if (buf[i]) (void)hipFree(buf[i]);
			buf[i] = 0;
			cap[i] = 0;
		}
	}

	// ------------------------------------------------------------------------------------------ the pass
	namespace
	{
		enum { SEND_R, SEND_C, RECV_R, RECV_C, OVF_MINE, OVF_ALL, SID, SID2, ADDR, OWNER, MISC_A, MISC_B, PACKED, REC, REC2, GATHER, SEND_R2, SEND_C2,
		       CMB_PAYLOAD, CMB_DIR, CMB_SEND, CMB_RECV, CMB_DIRS, CMB_MERGED };

		void Exchange(ShardedRank & r, Transport & net, int pass, const uint64_t * geom, void * sendR, void * sendC, uint64_t overflow, bool overflowFetched);

		// hash -> exchange -> (overflow lists) ; returns the receive buffers in r.buf[RECV_R], r.buf[RECV_C]
		void HashAndExchange(ShardedRank & r, Transport & net, int pass, const uint64_t * geom, uint64_t batch, uint64_t lo, uint64_t hi)
		{
			const int W = net.Ranks();
			void * sendR = r.Ensure(SEND_R, size_t(W) * geom[2]);
			void * sendC = r.Ensure(SEND_C, size_t(W) * geom[3]);
			uint64_t overflow = 0;
			r.Phase("buffers");
			LibCheck(r.ctx, tpc_shard_hash(r.ctx, pass, batch, lo, hi, sendR, sendC, &overflow), "shard_hash");
			r.Phase(pass == TPC_SHARD_INSERT ? "insert hash" : "query hash");
			Exchange(r, net, pass, geom, sendR, sendC, overflow, false);
		}

		// This rank's produced overflow entries of the batch just hashed, copied aside (r.buf[OVF_MINE]) before another hash of the
		// same pass may append to the list (TWOPACO_OVERLAP: the next batch's hash runs under this batch's exchange).
		void FetchOverflow(ShardedRank & r, int pass, const uint64_t * geom, uint64_t overflow)
		{
			if (overflow == 0 || overflow >= (uint64_t(1) << 62)) return;
			char * mine = static_cast<char*>(r.Ensure(OVF_MINE, overflow * geom[6]));
			LibCheck(r.ctx, tpc_shard_overflow_get(r.ctx, pass, mine, overflow), "shard_overflow_get");
		}

		// levels 2-3 over what Exchange left in r.buf[RECV_R] / [RECV_C] (and, unpacked, this rank's own block in the send buffers)
		void Apply(ShardedRank & r, int pass, uint64_t batch, const void * sendR, const void * sendC, uint64_t * survivors)
		{
			if (r.compactExchange) LibCheck(r.ctx, tpc_shard_apply_packed(r.ctx, pass, batch, r.buf[RECV_R], r.buf[RECV_C], survivors), "shard_apply_packed");
			else LibCheck(r.ctx, tpc_shard_apply_inplace(r.ctx, pass, batch, r.buf[RECV_R], r.buf[RECV_C], sendR, sendC, survivors), "shard_apply_inplace");
		}

		// everything between a batch's hash and its apply: counts, (packed) regions, overflow lists
		void Exchange(ShardedRank & r, Transport & net, int pass, const uint64_t * geom, void * sendR, void * sendC, uint64_t overflow, bool overflowFetched)
		{
			const int W = net.Ranks();
			// (one rank, unpacked: nothing moves and tpc_shard_apply_inplace reads the send buffers -- no receive buffer for the regions)
			void * recvR = (W > 1 || r.compactExchange) ? r.Ensure(RECV_R, size_t(W) * geom[2]) : 0;
			void * recvC = r.Ensure(RECV_C, size_t(W) * geom[3]);
			net.AllToAll(r.rank, sendC, recvC, geom[3]);
			r.Phase("exchange counts");
			if (r.compactExchange)
			{
				// the fixed-capacity regions are about 3/4 full: pack their used prefixes and move exactly those
				// (every destination's share is a whole number of 128-byte lines)
				void * packed = r.Ensure(PACKED, size_t(W) * geom[2]);
				std::vector<uint64_t> bytes(W), all;
				r.Phase("buffers");
				LibCheck(r.ctx, tpc_shard_pack(r.ctx, pass, sendR, sendC, packed, bytes.data()), "shard_pack");
				r.Phase("pack");
				net.ExchangeHost(r.rank, bytes.data(), W, all);
				std::vector<uint64_t> sendUnits(W), recvUnits(W);
				for (int s = 0; s < W; s++)
				{
					sendUnits[s] = bytes[s] / 16;
					recvUnits[s] = all[size_t(s) * W + r.rank] / 16;
					if (s != r.rank) r.regionBytesSent += bytes[s];  // what leaves this rank
				}

				net.AllToAllV(r.rank, packed, sendUnits.data(), recvR, recvUnits.data(), 16);
				r.Phase(pass == TPC_SHARD_INSERT ? "insert all-to-all" : "query all-to-all");
			}
			else
			{
				// the equal blocks as they are (tight regions: tpc_shard_plan); this rank's own block stays where it was hashed
				net.AllToAll(r.rank, sendR, recvR, geom[2], true);
				r.regionBytesSent += uint64_t(W - 1) * geom[2];
				r.Phase(pass == TPC_SHARD_INSERT ? "insert all-to-all" : "query all-to-all");
			}

			// skew path: entries that did not fit their level-1 region travel as one all-gathered list
			std::vector<uint64_t> all;
			net.ExchangeHost(r.rank, &overflow, 1, all);
			const uint64_t most = *std::max_element(all.begin(), all.end());
			if (most >= (uint64_t(1) << 62))
			{
				throw std::runtime_error("address-sharded pass: an overflow list overflowed (adversarial address skew)");
			}

			if (most > 0)
			{
				const size_t eb = geom[6];
				char * mine = static_cast<char*>(r.buf[OVF_MINE]);
				if (!overflowFetched || r.cap[OVF_MINE] < most * eb)
				{
					// (fetched early: the entries must survive the buffer growing to the all-gather's block size)
					char * grown = 0;
					if (overflowFetched && overflow)
					{
						HipCheck(hipMalloc(reinterpret_cast<void**>(&grown), most * eb), "overflow block");
						HipCheck(hipMemcpy(grown, mine, overflow * eb, hipMemcpyDeviceToDevice), "overflow block");
					}

					mine = static_cast<char*>(r.Ensure(OVF_MINE, most * eb));
					if (grown)
					{
						HipCheck(hipMemcpy(mine, grown, overflow * eb, hipMemcpyDeviceToDevice), "overflow block");
						(void)hipFree(grown);
					}
					else if (!overflowFetched) LibCheck(r.ctx, tpc_shard_overflow_get(r.ctx, pass, mine, overflow), "shard_overflow_get");
				}

				// the all-gathered blocks (each holds all[s] valid entries) in the first half, their valid prefixes back to back in the
				// second (a compaction in place would copy between overlapping ranges)
				char * gathered = static_cast<char*>(r.Ensure(OVF_ALL, 2 * size_t(W) * most * eb));
				char * list = gathered + size_t(W) * most * eb;
				net.AllGather(r.rank, mine, gathered, most * eb);
				uint64_t total = 0;
				for (int s = 0; s < W; s++)
				{
					if (all[s]) HipCheck(hipMemcpy(list + total * eb, gathered + size_t(s) * most * eb, all[s] * eb, hipMemcpyDeviceToDevice), "overflow compaction");
					total += all[s];
				}

				LibCheck(r.ctx, tpc_shard_overflow_set(r.ctx, pass, list, total), "shard_overflow_set");
			}

			r.Phase("overflow lists");
		}

		// The survivors of the last tpc_shard_apply go back to the rank that hashed their position -- it rides in the id -- so that they
		// are verified where their text is (the library groups them by that rank).  Returns how many arrived here (ids in r.buf[SID]).
		uint64_t ReturnSurvivors(ShardedRank & r, Transport & net, uint64_t n)
		{
			const int W = net.Ranks();
			uint64_t * tmp = W > 1 ? static_cast<uint64_t*>(r.Ensure(MISC_A, std::max<uint64_t>(n, 1) * 8)) : 0;
			uint64_t * send = static_cast<uint64_t*>(r.Ensure(MISC_B, std::max<uint64_t>(n, 1) * 8));
			uint64_t counts[64];
			LibCheck(r.ctx, tpc_shard_survivors_home(r.ctx, tmp, send, counts), "shard_survivors_home");
			if (W == 1)
			{
				std::swap(r.buf[SID], r.buf[MISC_B]);
				std::swap(r.cap[SID], r.cap[MISC_B]);
				return n;
			}

			std::vector<uint64_t> all;
			net.ExchangeHost(r.rank, counts, W, all);
			std::vector<uint64_t> recvCounts(W);
			uint64_t arriving = 0;
			for (int s = 0; s < W; s++)
			{
				recvCounts[s] = all[size_t(s) * W + r.rank];
				arriving += recvCounts[s];
			}

			uint64_t * mine = static_cast<uint64_t*>(r.Ensure(SID, std::max<uint64_t>(arriving, 1) * 8));
			net.AllToAllV(r.rank, send, counts, mine, recvCounts.data(), 8);
			return arriving;
		}

		// Survivors (ids in r.buf[SID]) against functions fn .. fn+count-1; returns the number that passed.  last: they are marked
		// (tpc_shard_finish); otherwise their ids are left in r.buf[SID] for the next step.
		uint64_t VerifyStep(ShardedRank & r, Transport & net, uint64_t n, int fn, int count, bool last)
		{
			const int W = net.Ranks();
			const uint64_t items = n * uint64_t(count);
			uint64_t * sid = static_cast<uint64_t*>(r.buf[SID]);
			// probe addresses in owner-major send order and the slot of every probe, from the library (one rank: the natural order)
			uint64_t * tmp = W > 1 ? static_cast<uint64_t*>(r.Ensure(ADDR, std::max<uint64_t>(items, 1) * 8)) : 0;
			uint32_t * perm = W > 1 ? static_cast<uint32_t*>(r.Ensure(MISC_A, std::max<uint64_t>(items, 1) * 4)) : 0;
			uint64_t * sendAddr = static_cast<uint64_t*>(r.Ensure(MISC_B, std::max<uint64_t>(items, 1) * 8));
			uint64_t counts[64];
			LibCheck(r.ctx, tpc_shard_verify_send(r.ctx, fn, count, sid, n, tmp, sendAddr, perm, counts), "shard_verify_send");
			uint8_t * back = 0;
			if (W == 1)
			{
				back = static_cast<uint8_t*>(r.Ensure(OWNER, std::max<uint64_t>(items, 1)));
				LibCheck(r.ctx, tpc_shard_probe(r.ctx, sendAddr, items, back), "shard_probe");
			}
			else
			{
				std::vector<uint64_t> all;
				net.ExchangeHost(r.rank, counts, W, all);
				std::vector<uint64_t> recvCounts(W);
				uint64_t asked = 0;
				for (int s = 0; s < W; s++)
				{
					recvCounts[s] = all[size_t(s) * W + r.rank];
					asked += recvCounts[s];
				}

				// requests in (ADDR: the tagged addresses are no longer needed), answers out
				uint64_t * req = static_cast<uint64_t*>(r.Ensure(ADDR, std::max<uint64_t>(std::max(asked, items), 1) * 8));
				net.AllToAllV(r.rank, sendAddr, counts, req, recvCounts.data(), 8);
				uint8_t * answers = static_cast<uint8_t*>(r.Ensure(OWNER, std::max<uint64_t>(std::max(asked, items * 4), 1)));
				LibCheck(r.ctx, tpc_shard_probe(r.ctx, req, asked, answers), "shard_probe");
				back = static_cast<uint8_t*>(r.Ensure(MISC_B, std::max<uint64_t>(items * 8, 1)));
				net.AllToAllV(r.rank, answers, recvCounts.data(), back, counts, 1);
			}

			uint64_t m = 0;
			if (last)
			{
				LibCheck(r.ctx, tpc_shard_finish(r.ctx, sid, n, count, back, perm, &m), "shard_finish");
				return m;
			}

			uint64_t * kept = static_cast<uint64_t*>(r.Ensure(SID2, std::max<uint64_t>(n, 1) * 8));
			LibCheck(r.ctx, tpc_shard_select(r.ctx, sid, n, count, back, perm, kept, &m), "shard_select");
			std::swap(r.buf[SID], r.buf[SID2]);
			std::swap(r.cap[SID], r.cap[SID2]);
			return m;
		}
	}

	namespace
	{
		// survivors of the first probe of one query batch: home, verified, marked (shared by both forms of the pass)
		void VerifyBatch(ShardedRank & r, Transport & net, int hashFunctions, uint64_t n);
		void MaskUnion(ShardedRank & r, Transport & net);

		// TWOPACO_OVERLAP=1: the same pass with every level-1 hash but the first running on the context's second stream under the
		// exchange and the apply of the batch before it (tpc_shard_hash_begin / _end, two send-buffer pairs) -- and the query's FIRST
		// hash under the insert's LAST exchange: the query's level 1 does not depend on the filter, only its lookup does (the
		// reference's workers never wait on each other for the filter either, vertexenumerator.h:1086-1092).  Same bytes on the
		// wire, same filter, same marks (tests/test_gpu_multigpu_host.py); what it buys on real links is for the first multi-GPU
		// box to say -- the hash kernels hold whole CUs (1024 threads, ~150 KB of LDS), so RCCL's send / receive workgroups
		// compete with them for CUs.
		void OverlappedFirstPass(ShardedRank & r, Transport & net, int hashFunctions, uint64_t lo, uint64_t hi)
		{
			const int W = net.Ranks();
			uint64_t gi[16], gq[16];
			r.PhaseBegin();
			LibCheck(r.ctx, tpc_shard_plan_both(r.ctx, lo, hi, gi, gq), "shard_plan_both");
			LibCheck(r.ctx, tpc_filter_reset(r.ctx), "filter_reset");
			r.Phase("plan");
			int slot = 0;
			auto sendBuffers = [&](int which, const uint64_t * geom, void * & R, void * & C)
			{
				R = r.Ensure(which ? SEND_R2 : SEND_R, size_t(W) * geom[2]);
				C = r.Ensure(which ? SEND_C2 : SEND_C, size_t(W) * geom[3]);
			};

			void * curR = 0, * curC = 0, * nextR = 0, * nextC = 0;
			uint64_t overflow = 0;
			sendBuffers(slot, gi, curR, curC);
			LibCheck(r.ctx, tpc_shard_hash(r.ctx, TPC_SHARD_INSERT, 0, lo, hi, curR, curC, &overflow), "shard_hash");
			r.Phase("insert hash");
			for (uint64_t b = 0; b < gi[0]; b++)
			{
				FetchOverflow(r, TPC_SHARD_INSERT, gi, overflow);
				const bool more = b + 1 < gi[0];
				sendBuffers(slot ^ 1, more ? gi : gq, nextR, nextC);
				if (more) LibCheck(r.ctx, tpc_shard_hash_begin(r.ctx, TPC_SHARD_INSERT, b + 1, lo, hi, nextR, nextC), "shard_hash_begin");
				else LibCheck(r.ctx, tpc_shard_hash_begin(r.ctx, TPC_SHARD_QUERY, 0, lo, hi, nextR, nextC), "shard_hash_begin(query)");
				Exchange(r, net, TPC_SHARD_INSERT, gi, curR, curC, overflow, true);
				Apply(r, TPC_SHARD_INSERT, b, curR, curC, 0);
				r.Phase("insert apply");
				if (more) LibCheck(r.ctx, tpc_shard_hash_end(r.ctx, TPC_SHARD_INSERT, &overflow), "shard_hash_end");
				slot ^= 1;
				curR = nextR; curC = nextC;
			}

			net.Barrier().Wait();  // every shard is complete before anyone probes it
			r.Phase("barrier");
			LibCheck(r.ctx, tpc_shard_hash_end(r.ctx, TPC_SHARD_QUERY, &overflow), "shard_hash_end(query)");
			r.Phase("query hash (under the insert)");
			for (uint64_t b = 0; b < gq[0]; b++)
			{
				FetchOverflow(r, TPC_SHARD_QUERY, gq, overflow);
				const bool more = b + 1 < gq[0];
				if (more)
				{
					sendBuffers(slot ^ 1, gq, nextR, nextC);
					LibCheck(r.ctx, tpc_shard_hash_begin(r.ctx, TPC_SHARD_QUERY, b + 1, lo, hi, nextR, nextC), "shard_hash_begin");
				}

				Exchange(r, net, TPC_SHARD_QUERY, gq, curR, curC, overflow, true);
				uint64_t n = 0;
				Apply(r, TPC_SHARD_QUERY, b, curR, curC, &n);
				r.Phase("query apply");
				VerifyBatch(r, net, hashFunctions, n);
				if (more) LibCheck(r.ctx, tpc_shard_hash_end(r.ctx, TPC_SHARD_QUERY, &overflow), "shard_hash_end");
				slot ^= 1;
				curR = nextR; curC = nextC;
			}

			r.PhasePrint("sharded first pass (hashes overlapped)");
		}
	}

	namespace
	{
		// The first pass with the filter replicated through set-bit lists (multigpu.h: ShardedRank::combined; the same protocol and the
		// same arithmetic -- tpc_combine_choose -- as twopaco_amd/dist.py:Combined).  The shared ConcurrentBitVector of the reference's
		// threads (concurrentbitvector.cpp:31-45, MergeOr :115-122) becomes: local insert -> export of every slice's set bits -> all-gather,
		// or reduce-scatter by owner + merge + all-gather (or the dense filters OR-reduced by word ranges) -> import -> local query.
		void DenseReduce(ShardedRank & r, Transport & net)
		{
			const int W = net.Ranks();
			if (W == 1) return;
			const uint64_t words = tpc_filter_words(r.ctx) - 1;  // 2^L / 32: a multiple of the number of ranks
			const uint64_t chunk = words / uint64_t(W);
			uint32_t * mine = static_cast<uint32_t*>(r.Ensure(CMB_SEND, words * 4));
			uint32_t * parts = static_cast<uint32_t*>(r.Ensure(CMB_RECV, words * 4));
			uint32_t * folded = static_cast<uint32_t*>(r.Ensure(CMB_MERGED, chunk * 4));
			LibCheck(r.ctx, tpc_filter_copy_out(r.ctx, 0, words, mine), "filter_copy_out");
			net.AllToAll(r.rank, mine, parts, chunk * 4);
			LibCheck(r.ctx, tpc_mask_or_blocks(r.ctx, parts, uint32_t(W), chunk, folded), "mask_or_blocks");
			net.AllGather(r.rank, folded, mine, chunk * 4);
			LibCheck(r.ctx, tpc_filter_copy_in(r.ctx, 0, words, mine), "filter_copy_in");
			r.combineBytesReceived = 2 * uint64_t(W - 1) * chunk * 4;
			r.combineMode = "dense";
		}

		void CombinedInsert(ShardedRank & r, Transport & net, uint64_t lo, uint64_t hi)
		{
			const int W = net.Ranks();
			LibCheck(r.ctx, tpc_filter_reset(r.ctx), "filter_reset");
			LibCheck(r.ctx, tpc_pass1_insert(r.ctx, lo, hi, 0), "pass1_insert");
			r.Phase("insert (local)");
			uint64_t info[8];
			LibCheck(r.ctx, tpc_combine_info(r.ctx, uint32_t(W), info), "combine_info");
			// every rank takes the same road: lists only if every rank's insert stayed in its level-2 regions
			std::vector<uint64_t> all;
			const char * pin = std::getenv("TWOPACO_COMBINE");  // gather | scatter | dense (measurements, tests)
			uint64_t sparse = info[0] && !(pin && std::string(pin) == "dense") ? 1 : 0;
			net.ExchangeHost(r.rank, &sparse, 1, all);
			for (int s = 0; s < W; s++) sparse = std::min<uint64_t>(sparse, all[size_t(s)]);
			if (!sparse)
			{
				DenseReduce(r, net);
				r.Phase("insert exchange (dense)");
				return;
			}

			const uint64_t nWin = info[2], slices = info[1], spd = slices / uint64_t(W), cap = info[3];
			uint16_t * payload = static_cast<uint16_t*>(r.Ensure(CMB_PAYLOAD, size_t(W) * cap * 16));
			uint64_t * dir = static_cast<uint64_t*>(r.Ensure(CMB_DIR, size_t(slices) * nWin * 8));
			std::vector<uint64_t> units(W);
			LibCheck(r.ctx, tpc_combine_export(r.ctx, uint32_t(W), payload, cap, dir, units.data()), "combine_export");
			r.Phase("insert export");
			// the query's level-1 hash and level-2 binning do not read the filter: enqueued now, they run while the lists travel
			// (TWOPACO_COMBINE_OVERLAP=0: not)
			const char * ov = std::getenv("TWOPACO_COMBINE_OVERLAP");
			if (!(ov && ov[0] == '0')) LibCheck(r.ctx, tpc_pass1_query_begin(r.ctx, lo, hi), "pass1_query_begin");
			net.ExchangeHost(r.rank, units.data(), W, all);  // all[s * W + d]: units of rank s for destination d
			uint64_t total = 0, most = 0, mine = 0;
			for (int s = 0; s < W; s++)
			{
				uint64_t sum = 0;
				for (int d = 0; d < W; d++) sum += all[size_t(s) * W + d];
				total += sum;
				most = std::max(most, sum);
				if (s == r.rank) mine = sum;
			}

			double model[3] = { 0, 0, 0 };
			int mode = tpc_combine_choose(uint32_t(W), r.filterBits, total / uint64_t(W), model);
			if (pin && std::string(pin) == "gather") mode = 1;
			if (pin && std::string(pin) == "scatter") mode = 2;
			if (W == 1) mode = 1;
			if (r.rank == 0 && std::getenv("TWOPACO_TIMING"))
			{
				std::fprintf(stderr, "[timing]     combined exchange: export %.1f MB of set-bit lists per rank; bytes a rank receives by the model: all-gather %.1f MB, "
					"reduce-scatter + all-gather %.1f MB, dense %.1f MB -> %s\n", 16.0 * double(total) / W / 1e6, model[0] / 1e6, model[1] / 1e6, model[2] / 1e6,
					mode == 1 ? "all-gather" : mode == 2 ? "reduce-scatter + all-gather" : "dense");
			}

			// the used prefixes of this rank's W blocks, back to back
			char * send = static_cast<char*>(r.Ensure(CMB_SEND, std::max<uint64_t>(most, 1) * 16));
			{
				uint64_t o = 0;
				for (int d = 0; d < W; d++)
				{
					if (units[d]) HipCheck(hipMemcpy(send + o * 16, reinterpret_cast<const char*>(payload) + size_t(d) * cap * 16, units[d] * 16, hipMemcpyDeviceToDevice), "pack the export");
					o += units[d];
				}
			}

			if (mode == 3)
			{
				// the export took the insert out of its regions: this rank's own lists bring it back, to be applied to its dense filter
				std::vector<uint64_t> base(W);
				for (int d = 0; d < W; d++) base[d] = uint64_t(d) * cap;
				LibCheck(r.ctx, tpc_combine_import(r.ctx, uint32_t(W), uint32_t(W), payload, base.data(), dir, spd * nWin), "combine_import");
				DenseReduce(r, net);
				r.Phase("insert exchange (dense)");
				return;
			}

			if (mode == 1)
			{
				// all-gather of every rank's blocks and directories: W x W blocks, rank-major
				char * allp = static_cast<char*>(r.Ensure(CMB_RECV, size_t(W) * std::max<uint64_t>(most, 1) * 16));
				uint64_t * alld = static_cast<uint64_t*>(r.Ensure(CMB_DIRS, size_t(W) * slices * nWin * 8));
				net.AllGather(r.rank, send, allp, std::max<uint64_t>(most, 1) * 16);
				net.AllGather(r.rank, dir, alld, slices * nWin * 8);
				std::vector<uint64_t> base;
				for (int s = 0; s < W; s++)
				{
					uint64_t o = uint64_t(s) * std::max<uint64_t>(most, 1);
					for (int d = 0; d < W; d++) { base.push_back(o); o += all[size_t(s) * W + d]; }
				}

				LibCheck(r.ctx, tpc_combine_import(r.ctx, uint32_t(W * W), uint32_t(W), reinterpret_cast<const uint16_t*>(allp), base.data(), alld, spd * nWin), "combine_import");
				r.combineBytesReceived = 16 * (total - mine) + uint64_t(W - 1) * slices * nWin * 8;
				r.combineMode = "gather";
				r.Phase("insert exchange (all-gather)");
				return;
			}

			// reduce-scatter: block d and its directory to rank d, merged there, the merged lists all-gathered
			std::vector<uint64_t> recvUnits(W), base(W);
			uint64_t arriving = 0;
			for (int s = 0; s < W; s++)
			{
				recvUnits[s] = all[size_t(s) * W + r.rank];
				base[s] = arriving;
				arriving += recvUnits[s];
			}

			char * recv = static_cast<char*>(r.Ensure(CMB_RECV, std::max<uint64_t>(arriving, 1) * 16));
			uint64_t * rdir = static_cast<uint64_t*>(r.Ensure(CMB_DIRS, size_t(W) * slices * nWin * 8));  // (room for the all-gathered merged directories below)
			net.AllToAllV(r.rank, send, units.data(), recv, recvUnits.data(), 16);
			net.AllToAll(r.rank, dir, rdir, spd * nWin * 8);
			r.Phase("insert exchange (reduce-scatter)");
			uint16_t * merged = static_cast<uint16_t*>(r.Ensure(CMB_MERGED, std::max<uint64_t>(arriving, 1) * 16));
			uint64_t mergedUnits = 0;
			LibCheck(r.ctx, tpc_combine_merge(r.ctx, uint32_t(W), reinterpret_cast<const uint16_t*>(recv), base.data(), rdir, merged, std::max<uint64_t>(arriving, 1), dir, &mergedUnits), "combine_merge");
			r.Phase("insert merge");
			net.ExchangeHost(r.rank, &mergedUnits, 1, all);
			uint64_t mostMerged = 1, totalMerged = 0;
			for (int s = 0; s < W; s++) { mostMerged = std::max(mostMerged, all[size_t(s)]); totalMerged += all[size_t(s)]; }
			// (the largest merged block sets the all-gather's block size: this rank's block padded in a buffer of that size)
			char * padded = static_cast<char*>(r.Ensure(CMB_SEND, mostMerged * 16));
			if (mergedUnits) HipCheck(hipMemcpy(padded, merged, mergedUnits * 16, hipMemcpyDeviceToDevice), "pad the merged block");
			char * allp = static_cast<char*>(r.Ensure(CMB_PAYLOAD, size_t(W) * mostMerged * 16));
			net.AllGather(r.rank, padded, allp, mostMerged * 16);
			net.AllGather(r.rank, dir, rdir, spd * nWin * 8);  // (dir now holds the merged lists' directory, [slice][window] of this rank's slices)
			for (int s = 0; s < W; s++) base[s] = uint64_t(s) * mostMerged;
			LibCheck(r.ctx, tpc_combine_import(r.ctx, uint32_t(W), uint32_t(W), reinterpret_cast<const uint16_t*>(allp), base.data(), rdir, spd * nWin), "combine_import");
			r.combineBytesReceived = 16 * (arriving - recvUnits[r.rank]) + 16 * (totalMerged - mergedUnits) + uint64_t(W - 1) * 2 * spd * nWin * 8;
			r.combineMode = "scatter";
			r.Phase("insert exchange (all-gather of the merged lists)");
		}

		void CombinedFirstPass(ShardedRank & r, Transport & net, uint64_t lo, uint64_t hi)
		{
			r.PhaseBegin();
			if (!r.filterLoaded) CombinedInsert(r, net, lo, hi);
			uint64_t marks = 0;
			LibCheck(r.ctx, tpc_pass1_query(r.ctx, lo, hi, &marks), "pass1_query");
			r.Phase("query (local)");
			if (r.phaseOn) std::fprintf(stderr, "[timing]     combined exchange: %s, %.1f MB received by rank 0 (%llu bytes)\n", r.combineMode.c_str(), double(r.combineBytesReceived) / 1e6,
				(unsigned long long)r.combineBytesReceived);
			r.PhasePrint("combined first pass");
		}
	}

	namespace
	{
		void MaskUnion(ShardedRank & r, Transport & net);
	}

	void ShardedFirstPass(ShardedRank & r, Transport & net, int hashFunctions, uint64_t lo, uint64_t hi)
	{
		if (r.combined)
		{
			CombinedFirstPass(r, net, lo, hi);
			if (!r.shardedSecondPass) MaskUnion(r, net);  // (every rank marked only the positions it hashed)
			return;
		}

		uint64_t geom[16];
		const bool overlapped = std::getenv("TWOPACO_OVERLAP") != 0 && !r.filterLoaded;
		if (overlapped) OverlappedFirstPass(r, net, hashFunctions, lo, hi);
		if (!overlapped) r.PhaseBegin();
		// ---- insert (FilterFillerWorker); skipped when the shard came from a checkpoint
		if (!r.filterLoaded && !overlapped)
		{
			LibCheck(r.ctx, tpc_shard_plan(r.ctx, TPC_SHARD_INSERT, lo, hi, geom), "shard_plan(insert)");
			LibCheck(r.ctx, tpc_filter_reset(r.ctx), "filter_reset");
			r.Phase("plan");
			for (uint64_t b = 0; b < geom[0]; b++)
			{
				HashAndExchange(r, net, TPC_SHARD_INSERT, geom, b, lo, hi);
				Apply(r, TPC_SHARD_INSERT, b, r.buf[SEND_R], r.buf[SEND_C], 0);
				r.Phase("insert apply");
			}
		}

		if (!overlapped)
		{
		net.Barrier().Wait();  // every shard is complete before anyone probes it
		r.Phase("barrier");
		// ---- query (CandidateCheckingWorker)
		LibCheck(r.ctx, tpc_shard_plan(r.ctx, TPC_SHARD_QUERY, lo, hi, geom), "shard_plan(query)");
		r.Phase("plan");
		for (uint64_t b = 0; b < geom[0]; b++)
		{
			HashAndExchange(r, net, TPC_SHARD_QUERY, geom, b, lo, hi);
			uint64_t n = 0;
			Apply(r, TPC_SHARD_QUERY, b, r.buf[SEND_R], r.buf[SEND_C], &n);
			r.Phase("query apply");
			VerifyBatch(r, net, hashFunctions, n);
		}

		r.PhasePrint("sharded first pass");
		}

		// positions inside homopolymer / dinucleotide tracts sent no probes (option shard_periodic_skip, set with the rank's context): they take
		// the verdict of the position whose window they repeat -- same tile, so same rank -- before anything reads the round mask
		LibCheck(r.ctx, tpc_shard_periodic_copy(r.ctx), "shard_periodic_copy");
		if (r.shardedSecondPass) return;  // the marks stay on the rank that found them (ShardedSecondPass)
		MaskUnion(r, net);
	}

	namespace
	{
		void VerifyBatch(ShardedRank & r, Transport & net, int hashFunctions, uint64_t n)
		{
			const int W = net.Ranks();
			n = ReturnSurvivors(r, net, n);
			r.Phase("survivors home");
			// Lazy: function 1 alone first (it rejects all but a fill-rate share of the Bloom false positives), then the rest together --
			// two round trips.  Where most first-probe survivors are true second edges instead (the 62-genome workload: 54 of 58 M
			// pass every probe) every survivor makes both trips, and one trip with all q - 1 addresses is cheaper: the share that
			// passed function 1 in the first lazy batch, summed over the ranks (they must issue the same collectives), decides for
			// every batch after it.  TWOPACO_VERIFY_ROUNDS = lazy | eager pins the choice.
			if (r.verifyEager < 0)
			{
				const char * pin = std::getenv("TWOPACO_VERIFY_ROUNDS");
				if (pin && std::string(pin) == "eager") r.verifyEager = 1;
				else if (pin && std::string(pin) == "lazy") r.verifyEager = 0;
			}

			if (hashFunctions == 2 || (hashFunctions > 2 && r.verifyEager == 1))
			{
				n = VerifyStep(r, net, n, 1, hashFunctions - 1, true);
			}
			else if (hashFunctions > 2)
			{
				const uint64_t before = n;
				n = VerifyStep(r, net, n, 1, 1, false);
				if (r.verifyEager < 0)
				{
					const uint64_t mine[2] = { n, before };
					std::vector<uint64_t> all;
					net.ExchangeHost(r.rank, mine, 2, all);
					uint64_t passed = 0, asked = 0;
					for (int s = 0; s < W; s++) { passed += all[size_t(s) * 2]; asked += all[size_t(s) * 2 + 1]; }
					if (asked > 0) r.verifyEager = passed * 2 > asked ? 1 : 0;
				}

				n = VerifyStep(r, net, n, 2, hashFunctions - 2, true);
			}

			if (hashFunctions < 2) LibCheck(r.ctx, tpc_shard_mark(r.ctx, static_cast<uint64_t*>(r.buf[SID]), n), "shard_mark");  // (the first probe was the only one)
			r.Phase("verify + mark");
		}

		// ---- union of the candidate masks: OR all-reduce by word ranges
		void MaskUnion(ShardedRank & r, Transport & net)
		{
		const int W = net.Ranks();
		const uint64_t words = tpc_mask_words(r.ctx);
		const uint64_t chunk = (words + W - 1) / W;
		uint32_t * mine = static_cast<uint32_t*>(r.Ensure(SEND_R, size_t(W) * chunk * 4));
		uint32_t * parts = static_cast<uint32_t*>(r.Ensure(RECV_R, size_t(W) * chunk * 4));
		uint32_t * folded = static_cast<uint32_t*>(r.Ensure(MISC_A, chunk * 4));
		LibCheck(r.ctx, tpc_mask_export_padded(r.ctx, mine, W * chunk), "mask_export_padded");
		net.AllToAll(r.rank, mine, parts, chunk * 4);
		LibCheck(r.ctx, tpc_mask_or_blocks(r.ctx, parts, uint32_t(W), chunk, folded), "mask_or_blocks");
		net.AllGather(r.rank, folded, mine, chunk * 4);
		LibCheck(r.ctx, tpc_mask_import(r.ctx, mine), "mask_import");
		}
	}

	void ShardedSecondPass(ShardedRank & r, Transport & net, uint64_t abundance, uint64_t counters[4])
	{
		const int W = net.Ranks();
		uint64_t n = 0;
		LibCheck(r.ctx, tpc_pass2_marks(r.ctx, &n), "pass2_marks");
		const int rowWords = tpc_key_words(r.ctx) + 1;
		const size_t rowBytes = size_t(rowWords) * 8;
		uint64_t * records = static_cast<uint64_t*>(r.Ensure(REC, std::max<uint64_t>(n, 1) * rowBytes));
		int32_t * owner = static_cast<int32_t*>(r.Ensure(OWNER, std::max<uint64_t>(n, 1) * 4));
		// combine before routing: the rank's own exact filter first, then one record per DISTINCT key (sets, "seen twice", count) instead
		// of one per marked position (TWOPACO_PASS2_AGGREGATE=0: the per-position records of rounds 3-5)
		const bool aggregate = !(std::getenv("TWOPACO_PASS2_AGGREGATE") && std::getenv("TWOPACO_PASS2_AGGREGATE")[0] == '0');
		uint64_t rows = n;
		if (aggregate) LibCheck(r.ctx, tpc_pass2_aggregate_records(r.ctx, uint32_t(W), abundance, records, owner, &rows), "pass2_aggregate_records");
		else LibCheck(r.ctx, tpc_pass2_mark_records(r.ctx, uint32_t(W), records, owner), "pass2_mark_records");
		uint32_t * perm = static_cast<uint32_t*>(r.Ensure(MISC_A, std::max<uint64_t>(rows, 1) * 4));
		uint64_t counts[64];
		LibCheck(r.ctx, tpc_shard_route(r.ctx, owner, rows, perm, counts), "shard_route");
		uint64_t * send = static_cast<uint64_t*>(r.Ensure(MISC_B, std::max<uint64_t>(rows, 1) * rowBytes));
		LibCheck(r.ctx, tpc_shard_permute_rows(r.ctx, records, perm, rows, rowWords, send), "shard_permute_rows");
		std::vector<uint64_t> all;
		net.ExchangeHost(r.rank, counts, W, all);
		std::vector<uint64_t> recvCounts(W);
		uint64_t arriving = 0;
		for (int s = 0; s < W; s++)
		{
			recvCounts[s] = all[size_t(s) * W + r.rank];
			arriving += recvCounts[s];
		}

		uint64_t * mine = static_cast<uint64_t*>(r.Ensure(REC2, std::max<uint64_t>(arriving, 1) * rowBytes));
		net.AllToAllV(r.rank, send, counts, mine, recvCounts.data(), rowBytes);
		uint64_t truePositives = 0, falsePositives = 0, tableSize = 0;
		if (aggregate) LibCheck(r.ctx, tpc_pass2_filter_aggregated(r.ctx, mine, arriving, abundance, &truePositives, &falsePositives, &tableSize), "pass2_filter_aggregated");
		else LibCheck(r.ctx, tpc_pass2_filter_records(r.ctx, mine, arriving, abundance, &truePositives, &falsePositives, &tableSize), "pass2_filter_records");
		counters[0] = truePositives; counters[1] = falsePositives; counters[2] = tableSize; counters[3] = n;
	}

	void ShardedFinish(ShardedRank & r, Transport & net, uint64_t * junctions, bool gatherOnRankZero)
	{
		const int W = net.Ranks();
		const size_t keyBytes = size_t(tpc_key_words(r.ctx)) * 8;
		// ---- all junction keys on every rank
		uint64_t mineKeys = 0;
		LibCheck(r.ctx, tpc_junction_keys_export(r.ctx, 0, 0, &mineKeys), "junction_keys_export");
		std::vector<uint64_t> all;
		net.ExchangeHost(r.rank, &mineKeys, 1, all);
		const uint64_t most = std::max<uint64_t>(1, *std::max_element(all.begin(), all.end()));
		uint64_t * block = static_cast<uint64_t*>(r.Ensure(MISC_A, most * keyBytes));
		uint64_t * gathered = static_cast<uint64_t*>(r.Ensure(GATHER, size_t(W) * most * keyBytes));
		LibCheck(r.ctx, tpc_junction_keys_export(r.ctx, block, most, &mineKeys), "junction_keys_export");
		net.AllGather(r.rank, block, gathered, most * keyBytes);
		for (int s = 0; s < W; s++)
		{
			LibCheck(r.ctx, tpc_junction_keys_import(r.ctx, reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(gathered) + size_t(s) * most * keyBytes), all[s], s > 0 ? 1 : 0),
				"junction_keys_import");
		}

		LibCheck(r.ctx, tpc_junctions_finalize(r.ctx, junctions), "junctions_finalize");
		// ---- ids of this rank's positions, then every list to rank 0 (rank order = position order)
		uint64_t marked = 0, valid = 0;
		LibCheck(r.ctx, tpc_emit(r.ctx, &marked, &valid), "emit");
		if (!gatherOnRankZero) return;  // every rank formats its own part of the stream (ShardedStream)
		uint64_t * g = static_cast<uint64_t*>(r.Ensure(REC, std::max<uint64_t>(marked, 1) * 8));
		int64_t * id = static_cast<int64_t*>(r.Ensure(REC2, std::max<uint64_t>(marked, 1) * 8));
		LibCheck(r.ctx, tpc_emit_export(r.ctx, g, id), "emit_export");
		net.ExchangeHost(r.rank, &marked, 1, all);
		std::vector<uint64_t> sendCounts(W, 0), recvCounts(W, 0);
		sendCounts[0] = marked;
		uint64_t total = 0;
		if (r.rank == 0)
		{
			for (int s = 0; s < W; s++)
			{
				recvCounts[s] = all[s];
				total += all[s];
			}
		}

		uint64_t * gAll = static_cast<uint64_t*>(r.Ensure(MISC_B, std::max<uint64_t>(total, 1) * 8));
		int64_t * idAll = static_cast<int64_t*>(r.Ensure(GATHER, std::max<uint64_t>(total, 1) * 8));
		net.AllToAllV(r.rank, g, sendCounts.data(), gAll, recvCounts.data(), 8);
		net.AllToAllV(r.rank, id, sendCounts.data(), idAll, recvCounts.data(), 8);
		if (r.rank == 0) LibCheck(r.ctx, tpc_emit_import(r.ctx, gAll, idAll, total), "emit_import");
	}

	void ShardedStream(ShardedRank & r, Transport & net, const std::vector<uint64_t> & recStart, const std::vector<uint64_t> & recLength, size_t k,
		uint64_t * firstByte, uint64_t * nBytes, uint64_t * records)
	{
		const int W = net.Ranks();
		const size_t n = recStart.size();
		uint64_t lo = 0, hi = 0;
		if (tpc_shard_chunk(r.ctx, &lo, &hi) != 0) throw std::runtime_error("shard_chunk failed");
		// what this rank's marks say about every sequence
		std::vector<uint64_t> mine(2 * n);
		std::vector<uint32_t> flags(n);
		LibCheck(r.ctx, tpc_emit_stream_partial(r.ctx, recStart.data(), recLength.data(), uint32_t(n), mine.data(), flags.data()), "emit_stream_partial");
		for (size_t s = 0; s < n; s++) mine[n + s] = flags[s];
		std::vector<std::vector<uint64_t> > all;
		net.GatherHost(r.rank, mine, all);
		// the whole stream's layout, the same on every rank (tpc_stream.hip: slot arithmetic)
		std::vector<uint32_t> gflags(n, 0);
		std::vector<uint64_t> eScan(n + 1, 0), sScan(n + 1, 0), before(n, 0);
		uint32_t rLast = 0;
		for (size_t s = 0; s < n; s++)
		{
			if (recLength[s] < k) { eScan[s + 1] = eScan[s]; sScan[s + 1] = sScan[s]; continue; }
			rLast = uint32_t(s);
			uint64_t total = 0;
			uint32_t f = 4u;
			for (int q = 0; q < W; q++)
			{
				if (q < r.rank) before[s] += all[size_t(q)][s];
				total += all[size_t(q)][s];
				f |= uint32_t(all[size_t(q)][n + s]);
			}

			const uint64_t stubs = ((f & 1u) ? 0u : 1u) + ((recLength[s] != k && !(f & 2u)) ? 1u : 0u);
			gflags[s] = f;
			eScan[s + 1] = eScan[s] + total + stubs;
			sScan[s + 1] = sScan[s] + stubs;
		}

		// my slots: my real-id records, the stubs and the separators that belong to positions of my chunk
		uint64_t slots = 0;
		for (size_t s = 0; s < n; s++)
		{
			slots += all[size_t(r.rank)][s];
			if (s < rLast)
			{
				const uint64_t at = recStart[s + 1] - 1;
				if (at >= lo && at < hi) ++slots;
			}

			if (!(gflags[s] & 4u)) continue;
			const uint64_t first = recStart[s], last = first + recLength[s] - k;
			if (!(gflags[s] & 1u) && first >= lo && first < hi) ++slots;
			if (recLength[s] != k && !(gflags[s] & 2u) && last >= lo && last < hi) ++slots;
		}

		std::vector<uint64_t> allSlots;
		net.ExchangeHost(r.rank, &slots, 1, allSlots);
		uint64_t slot0 = 0, totalSlots = 0;
		for (int q = 0; q < W; q++)
		{
			if (q < r.rank) slot0 += allSlots[size_t(q)];
			totalSlots += allSlots[size_t(q)];
		}

		const uint64_t allRecords = eScan[n];
		if (totalSlots != allRecords + (allRecords ? rLast : 0)) throw std::runtime_error("junction stream: the ranks' slots do not add up");
		uint64_t bytes = 0;
		LibCheck(r.ctx, tpc_emit_stream_part(r.ctx, recStart.data(), recLength.data(), uint32_t(n), gflags.data(), eScan.data(), sScan.data(), before.data(), rLast,
			lo, hi, slot0, allRecords ? slots : 0, &bytes), "emit_stream_part");
		*firstByte = slot0 * 12;
		*nBytes = bytes;
		*records = allRecords;
	}
}
