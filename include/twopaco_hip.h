/*
 * twopaco_hip.h -- C-ABI of the MI355X junction-enumeration library (libtwopaco_hip.so).
 *
 * This is the drop-in boundary for TwoPaCo's two-pass junction enumeration.  The reference
 * has no FFI: its operator boundary is the C++ factory TwoPaCo::CreateEnumerator
 * (reference src/graphconstructor/vertexenumerator.h:37-46) whose constructor
 * (vertexenumerator.h:122-466) runs the worker classes below on CPU threads.  Each entry
 * point here replaces one of those workers / data structures; twopaco_amd/host/ keeps the
 * CreateEnumerator signature and calls only this ABI.  Reference paths are relative to
 * /root/reference/src ; VE.h = graphconstructor/vertexenumerator.h.
 *
 * Conventions: plain C, opaque context, int status (0 = ok, <0 = error, text via
 * tpc_last_error).  The caller owns every host buffer; the library owns device memory
 * until tpc_ctx_destroy.  One host thread per context; calls are synchronous with respect
 * to their outputs.  Device pointers (the *_dev entry points) are raw HIP device addresses.
 *
 * Text model.  The reference streams each FASTA record as 'N' + bases + 'N' in overlapping
 * Tasks (VE.h:1108-1226).  The library works on the equivalent global text
 *     T = N rec0 N rec1 N ... rec(S-1) N
 * indexed by a global position g (uint64): base g is the 2-bit code (A0 C1 G2 T3,
 * dnachar.cpp:18-33) at bits 2*(g%32) of bases[g/32] -- the CompressedString layout
 * (compressedstring.h:188-195) -- and bit g%32 of nmask[g/32] is set when T[g] is 'N'
 * (any non-ACGT character, VE.h:1174, and the separators).  A "vertex position" g is the
 * k-mer window T[g..g+k); its sequence coordinate is g - rec_start.
 */
#ifndef TWOPACO_HIP_H_
#define TWOPACO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tpc_ctx tpc_ctx;

#define TPC_MAX_Q 64                   /* hash functions: 1..16 on the rolling kernels (direct and partitioned), 17..64 on closed-form direct kernels */
#define TPC_INVALID_VERTEX INT64_MAX   /* graphconstructor/common.cpp:5                  */

/* Kernel ids for tpc_kernel_ms (hipEvent-timed duration of the last launch of each). */
enum {
    TPC_K_FILTER_RESET = 0, /* ConcurrentBitVector ctor zeroing, concurrentbitvector.cpp:11-24 */
    TPC_K_INSERT = 1,       /* FilterFillerWorker, VE.h:995-1105                               */
    TPC_K_QUERY = 2,        /* CandidateCheckingWorker, VE.h:586-704                           */
    TPC_K_COMPACT = 3,      /* candidate mask -> position list (replaces candidate_<r>.tmp)    */
    TPC_K_FILTER2 = 4,      /* CandidateFinalFilteringWorker, VE.h:708-829                     */
    TPC_K_SCAN2 = 5,        /* TrueBifurcations, VE.h:1228-1256                                */
    TPC_K_SORT = 6,         /* BifurcationStorage::Init sort, bifurcationstorage.h:65          */
    TPC_K_EMIT = 7,         /* EdgeConstructionWorker id lookup, VE.h:927-958                  */
    TPC_K_SPLIT = 8,        /* InitialFilterFillerWorker, VE.h:503-583                         */
    TPC_K_SHARD_HASH = 9,   /* tpc_shard_hash: level 1 of a sharded pass                       */
    TPC_K_SHARD_APPLY = 10, /* tpc_shard_apply: levels 2-3 of a sharded pass                   */
    TPC_K_STREAM = 11,      /* tpc_emit_stream: FlushEdgeResults + JunctionPositionWriter bytes  */
    TPC_K_FUSED = 12,       /* deferred apply: k_q_split + k_apply_lookup (inside TPC_K_QUERY), or the apply alone when it was flushed */
    TPC_K_LOOKUP = 13,      /* the k_apply_lookup launch of TPC_K_FUSED alone (the one kernel insert and query share: its time is split by bytes) */
    TPC_K_COMBINE = 14,     /* tpc_combine_export / tpc_combine_merge: k_slice_combine                                        */
    TPC_K_COUNT = 15
};

/* Context on HIP device `device`.  Fails (non-zero) when no GPU / device is present:
 * there is no CPU fallback behind this ABI. */
int tpc_ctx_create(int device, tpc_ctx **out);
void tpc_ctx_destroy(tpc_ctx *ctx);
const char *tpc_last_error(const tpc_ctx *ctx);

/* Optional: make the runtime load every kernel code object now (an attribute query of one kernel per translation unit,
 * ~30 ms in all) instead of at the first real launch.  Neither call needs a stream or touches context state, so a one-shot
 * caller (the CLI) runs tpc_warmup on a second host thread while it allocates the filter and uploads the text; tpc_preload
 * is the same for a device that has no context yet. */
int tpc_preload(int device);
int tpc_warmup(tpc_ctx *ctx);
/* Optional: allocate the partition buffers of the first pass now, for a text of at most n_text_max positions (after
 * tpc_set_params, before tpc_seq_upload), instead of inside the first tpc_pass1_insert.  A one-shot caller (the CLI) knows
 * an upper bound of the text length from its input files' sizes and lets this run beside parsing: device allocations of
 * tens of GiB take tens of milliseconds (the reference allocates its filter up front too, vertexenumerator.h:258-259). */
int tpc_reserve(tpc_ctx *ctx, uint64_t n_text_max);

/* Hash parameters: vertex length k, filter bits L (filter has 2^L bits), q functions and
 * their character tables seed_table[q][5] (A,C,G,T,N) -- the only entries of
 * CharacterHash::hashvalues the path ever reads (characterhash.h:41-59).  Replaces
 * VertexRollingHashSeed (vertexrollinghash.h:13-52). */
int tpc_set_params(tpc_ctx *ctx, int k, int L, int q, const uint64_t *seed_table);

/* Parse-once replacement for DistributeTasks (VE.h:1108-1226): upload the packed global
 * text.  bases has ceil(n_text/32) uint64 words, nmask ceil(n_text/32) uint32 words;
 * T[0] and T[n_text-1] must be 'N'. */
int tpc_seq_upload(tpc_ctx *ctx, const uint64_t *bases, const uint32_t *nmask, uint64_t n_text);

/* Start a new enumeration over the uploaded text: forgets the junction keys, masks and round
 * state of the previous run (the reference builds a fresh VertexEnumerator per run, VE.h:122). */
int tpc_run_begin(tpc_ctx *ctx);

/* ConcurrentBitVector(2^L) construction = zero fill (concurrentbitvector.cpp:11-24, VE.h:257). */
int tpc_filter_reset(tpc_ctx *ctx);

/* First-pass insert, FilterFillerWorker (VE.h:995-1105): every canonical (k+1)-mer edge of
 * every N-free vertex (A/T dummy edges beside N), gated by the round's vertex-hash range
 * [lo,hi] inclusive (VE.h:1063-1073).  n_kmers (may be NULL) receives the number of vertex
 * positions hashed. */
int tpc_pass1_insert(tpc_ctx *ctx, uint64_t lo, uint64_t hi, uint64_t *n_kmers);

/* Split pass histogram, InitialFilterFillerWorker (VE.h:503-583): uses (and overwrites) the
 * filter as scratch; bins_host receives 2^24 counters (VE.h:471).  records: global start
 * and length of every dispatched record (len >= k), n_rec of them. */
int tpc_pass1_split_hist(tpc_ctx *ctx, const uint64_t *rec_start, const uint64_t *rec_len, uint32_t n_rec,
                         uint32_t *bins_host);

/* First-pass query, CandidateCheckingWorker (VE.h:586-704): sets this round's candidate
 * mask (bit g) and returns the number of marks ("Candidate marks count", VE.h:387). */
int tpc_pass1_query(tpc_ctx *ctx, uint64_t lo, uint64_t hi, uint64_t *n_marks);
/* The part of the query that does not read the filter -- level-1 hash and level-2 binning of the first tile batch (the reference's
 * CandidateCheckingWorker computes its hashes before it touches the filter too, VE.h:633-640) -- enqueued on the context's stream; the
 * call returns without waiting.  The tpc_pass1_query of the same range that follows continues from there.  Between the two the caller
 * may move the round's insert between ranks (tpc_combine_merge / _import): the lists travel while the probes are being binned.  Optional. */
int tpc_pass1_query_begin(tpc_ctx *ctx, uint64_t lo, uint64_t hi);

/* Second-pass exact filter over this round's marks, CandidateFinalFilteringWorker
 * (VE.h:708-829) + TrueBifurcations (VE.h:1228-1256): appends the round's junction keys,
 * ORs the round mask into the run-wide mask (MergeOr, VE.h:909-913). Counters as logged at
 * VE.h:384-386. */
int tpc_pass2_filter(tpc_ctx *ctx, uint64_t abundance, uint64_t *n_true, uint64_t *n_false, uint64_t *table_size);

/* The same filter with its table sharded by key hash over `world` ranks (multi-GPU runs whose candidate marks stay on the rank
 * that hashed them, twopaco_amd/dist.py): tpc_pass2_marks compacts this round's mask into the ordered list of marked positions
 * (*n_marks of them, kept for the output pass); tpc_pass2_mark_owners copies that list to pos_dev and writes the rank that owns
 * each position's canonical key -- every occurrence of a k-mer, on either strand, goes to one owner; the host layer routes the
 * positions (tpc_shard_route / tpc_shard_permute64, a variable all-to-all of 8 bytes per marked position) and
 * tpc_pass2_filter_positions runs the exact filter over the positions a rank received: complete (prev, next) sets and counts per
 * key, hence the reference's verdict; it appends this rank's junction keys (to be all-gathered:
 * tpc_junction_keys_export / _import) and merges the round mask as tpc_pass2_filter does.  Needs the whole text on the rank. */
int tpc_pass2_marks(tpc_ctx *ctx, uint64_t *n_marks);
int tpc_pass2_mark_owners(tpc_ctx *ctx, uint32_t world, uint64_t *pos_dev, int32_t *owner_dev);
int tpc_pass2_filter_positions(tpc_ctx *ctx, const uint64_t *pos_dev, uint64_t n, uint64_t abundance, uint64_t *n_true, uint64_t *n_false,
                               uint64_t *table_size);
/* Text-free variant of the above: what travels is a record of key_words + 1 uint64 per marked position -- the canonical key and
 * prev | next << 3 as its strand sees them (candidateoccurence.h:25-50) -- so the owner needs none of the text and every rank
 * can keep just its chunk (option text_window).  tpc_pass2_mark_records writes the records of this rank's marks and their
 * owners, the host layer routes the rows (tpc_shard_route + tpc_shard_permute_rows, variable all-to-all),
 * tpc_pass2_filter_records is the exact filter over the received records. */
int tpc_pass2_mark_records(tpc_ctx *ctx, uint32_t world, uint64_t *records_dev, int32_t *owner_dev);
int tpc_pass2_filter_records(tpc_ctx *ctx, const uint64_t *records_dev, uint64_t n, uint64_t abundance, uint64_t *n_true, uint64_t *n_false,
                             uint64_t *table_size);
/* Combine before routing, second pass: tpc_pass2_aggregate_records runs the exact filter over this rank's OWN marks first (into its
 * table: CandidateFinalFilteringWorker's map per rank, VE.h:708-829) and writes one record per DISTINCT key -- the key and, in the last
 * word, bit 63 | the letter sets of prev / next, "seen at least twice" and the occurrence count its marks add up to -- with the key's
 * owner; *n_records of them (<= the rank's marks: size the buffers for those).  On many-genome inputs a key is marked dozens of times, so
 * far fewer rows travel (M2 at two ranks: 22 M marks -> ~1 M records per rank).  The rows are routed as before and
 * tpc_pass2_filter_aggregated merges what a rank received: sets OR-ed, counts added, the reference's verdict (isBif, abundance cut) on
 * the sums.  Occurrences are counted whenever `abundance` is a real cut (< 2^40) -- pass the same value to both calls on every rank.
 * tpc_pass2_filter_records takes aggregated and per-position records alike; what differs is how the table is sized and that rule. */
int tpc_pass2_aggregate_records(tpc_ctx *ctx, uint32_t world, uint64_t abundance, uint64_t *records_dev, int32_t *owner_dev, uint64_t *n_records);
int tpc_pass2_filter_aggregated(tpc_ctx *ctx, const uint64_t *records_dev, uint64_t n, uint64_t abundance, uint64_t *n_true, uint64_t *n_false,
                                uint64_t *table_size);
int tpc_shard_permute_rows(tpc_ctx *ctx, const uint64_t *src_dev, const uint32_t *perm_dev, uint64_t n, int row_words, uint64_t *dst_dev);

/* BifurcationStorage::Init (bifurcationstorage.h:27-66): sort all junction keys in
 * CompressedString::Less order (compressedstring.h:93-104) and build the id index. */
int tpc_junctions_finalize(tpc_ctx *ctx, uint64_t *n_junctions);

/* Capacity in 64-bit words of one key: CalculateNeededCapacity (candidateoccurence.h:129-133). */
int tpc_key_words(const tpc_ctx *ctx);

/* Sorted junction keys, n_junctions x key_words uint64 (what bifurcations.bin holds, sorted). */
int tpc_junction_keys(tpc_ctx *ctx, uint64_t *keys_host);

/* Junction keys appended so far by tpc_pass2_filter, unsorted (the content of the reference's
 * bifurcations.bin scratch file, VE.h:219,1228-1256): *n receives their number; keys_host (may be
 * NULL to query n only) receives n x key_words words.  tpc_junction_keys_set replaces the set --
 * the multi-GPU driver uses the pair to union the per-rank sets before tpc_junctions_finalize. */
int tpc_junction_keys_raw(tpc_ctx *ctx, uint64_t *keys_host, uint64_t *n);
int tpc_junction_keys_set(tpc_ctx *ctx, const uint64_t *keys_host, uint64_t n);
/* The same pair on DEVICE buffers, so that the union of the per-rank key sets can be all-gathered without
 * touching the host: _export copies min(n, cap_keys) keys to dst_dev and reports n; _import replaces
 * (append = 0) or extends (append = 1) the set with n keys from src_dev. */
int tpc_junction_keys_export(tpc_ctx *ctx, uint64_t *dst_dev, uint64_t cap_keys, uint64_t *n);
int tpc_junction_keys_import(tpc_ctx *ctx, const uint64_t *src_dev, uint64_t n, int append);

/* BifurcationStorage::GetId (bifurcationstorage.h:100-127) for one k-mer given as k ASCII
 * characters: +(rank+1), -(rank+1) or TPC_INVALID_VERTEX.  Host-side binary search over the
 * downloaded keys (VertexEnumerator::GetId is a cold query API, VE.h:99-102). */
int64_t tpc_get_id(tpc_ctx *ctx, const char *kmer);

/* Output pass id lookup, EdgeConstructionWorker (VE.h:927-940): for every marked N-free
 * position of the run-wide mask, in increasing g, its junction id (or TPC_INVALID_VERTEX for
 * a Bloom false positive).  Results stay in device memory; n_marked = list length, n_valid =
 * entries with a real id. */
int tpc_emit(tpc_ctx *ctx, uint64_t *n_marked, uint64_t *n_valid);
/* Copy the emit lists to the host: g_host[n_marked], id_host[n_marked]. */
int tpc_emit_fetch(tpc_ctx *ctx, uint64_t *g_host, int64_t *id_host);

/* The (position, id) lists tpc_emit left on the device, out to / in from device buffers: a multi-GPU host whose ranks each looked up
 * the ids of their own marked positions gathers the lists (in rank order = position order) on the rank that formats the
 * output and installs them there before tpc_emit_stream. */
int tpc_emit_export(tpc_ctx *ctx, uint64_t *g_dev, int64_t *id_dev);
int tpc_emit_import(tpc_ctx *ctx, const uint64_t *g_dev, const int64_t *id_dev, uint64_t n);

/* The output file's bytes, built on the device after tpc_emit: FlushEdgeResults (VE.h:837-854) +
 * JunctionPositionWriter::WriteJunction (junctionapi.h:118-132).  12-byte little-endian records
 * (u32 position in its sequence, i64 id) in (sequence, position) order; the first / last k-mer of
 * every sequence of >= k bases without a junction id gets a stub id n_junctions + 42, + 43, ...
 * (VE.h:419, 942-948); one separator (0xFFFFFFFF, INT64_MAX) per sequence-id step before the first
 * record of a sequence (junctionapi.h:120-123).  rec_start / rec_len: global text position and
 * length of EVERY input sequence (n_rec of them, short ones included: they consume an id).
 * n_bytes = stream length, n_records = records without separators ("True marks count", VE.h:451).
 * tpc_emit_stream_fetch copies [offset, offset + nbytes) to the host and may be called from several
 * threads at once (each into its own buffer; pinned buffers from tpc_host_alloc copy fastest). */
int tpc_emit_stream(tpc_ctx *ctx, const uint64_t *rec_start, const uint64_t *rec_len, uint32_t n_rec, uint64_t *n_bytes, uint64_t *n_records);
int tpc_emit_stream_fetch(tpc_ctx *ctx, uint64_t offset, uint64_t nbytes, void *dst_host);

/* The junction stream cut over the ranks of a multi-GPU run (each holds the marks and ids of its own chunk of the text after
 * tpc_emit): every rank formats and writes its own byte range of the file, nobody gathers (position, id) lists.  The
 * reference's ordered flush (FlushEdgeResults, VE.h:837-854; JunctionPositionWriter, junctionapi.h:118-126) becomes an
 * exclusive scan over per-sequence counts.  Every slot of the stream belongs to a text position (a record to its k-mer, a
 * stub to the end k-mer it stands for, the separator between sequences j and j + 1 to the character in front of j + 1), so
 * the slots of a chunk [chunk_lo, chunk_hi) are contiguous in the file.
 *   tpc_shard_chunk          this rank's chunk of text positions (the tiles it hashes; the last rank's end is UINT64_MAX)
 *   tpc_emit_stream_partial  per sequence: cnt_host[r] = real-id records among THIS rank's marks, flags_host[r] bit 0 / 1 =
 *                            this rank holds the first / last k-mer of the sequence with a real id
 *   (the host adds the ranks up: gflags = OR of the flags | 4 for sequences of >= k bases; per sequence n = sum of cnt + stubs,
 *    e_scan / s_scan = exclusive scans of n / stubs over the sequences (n_rec + 1 entries); before[r] = cnt of the ranks before
 *    this one; r_last = last sequence of >= k bases; slot0 = slots of the ranks before this one)
 *   tpc_emit_stream_part     formats this rank's n_slots slots (12 bytes each) into the stream buffer; fetch with
 *                            tpc_emit_stream_fetch (offsets relative to the rank's first byte) */
int tpc_shard_chunk(const tpc_ctx *ctx, uint64_t *chunk_lo, uint64_t *chunk_hi);
int tpc_emit_stream_partial(tpc_ctx *ctx, const uint64_t *rec_start, const uint64_t *rec_len, uint32_t n_rec, uint64_t *cnt_host, uint32_t *flags_host);
int tpc_emit_stream_part(tpc_ctx *ctx, const uint64_t *rec_start, const uint64_t *rec_len, uint32_t n_rec, const uint32_t *gflags_host,
                         const uint64_t *e_scan_host, const uint64_t *s_scan_host, const uint64_t *before_host, uint32_t r_last,
                         uint64_t chunk_lo, uint64_t chunk_hi, uint64_t slot0, uint64_t n_slots, uint64_t *n_bytes);
int tpc_host_alloc(void **ptr, uint64_t bytes);
void tpc_host_free(void *ptr);

/* One rank (world == 1, e.g. the sharded protocol exercised on a single device): the survivors of the last tpc_shard_apply(QUERY)
 * are verified against hash functions 1..q-1 and marked in place -- every survivor is home and every probe address is owned here,
 * so none of the routing calls below is needed (reference: CandidateCheckingWorker's remaining probes, vertexenumerator.h:640-660). */
int tpc_shard_verify_local(tpc_ctx *ctx);

/* Periodic windows under sharding.  A position whose k + 2 characters repeat those of the position 1 .. 63 before it (homopolymers,
 * microsatellites, telomeres) would send the same probes and the same insert as that position (CandidateCheckingWorker / FilterFillerWorker see
 * the same window: vertexenumerator.h:633-674, 1035-1092); the one-GPU passes skip such positions and copy the verdict afterwards.  A host
 * of the sharded calls opts in with tpc_set_option(ctx, "shard_periodic_skip", 1) before tpc_shard_plan -- tpc_shard_hash then skips them
 * too -- and MUST call tpc_shard_periodic_copy once per round, after the marks of the round's last query batch (tpc_shard_finish /
 * tpc_shard_mark / tpc_shard_verify_local) and before anything reads the round mask (mask union, tpc_pass2_marks).  The source of a copy
 * lies in the same 512-word tile, i.e. on the same rank.  Without the option nothing is skipped and the call does nothing. */
int tpc_shard_periodic_copy(tpc_ctx *ctx);

/* ---- address-sharded filter (multi-GPU) -------------------------------------------------
 * The Bloom filter (ConcurrentBitVector bitVector, VE.h:257) is cut over `world` ranks (a power of
 * two) by bit address: the partitioned passes route every address to the workgroup that owns its
 * filter slice, and rank r owns the slices of the level-1 buckets b1 with b1 % world == r.  Rank r
 * hashes the r-th contiguous chunk of the text's tiles (and, with option text_window, holds only that
 * chunk + halo); the level-1 regions are what travels (one equal-split all_to_all per pass and batch).  The library does no communication:
 * the caller (twopaco_amd/dist.py over torch.distributed, or MPI/RCCL in a C++ host) moves the
 * DEVICE buffers named below between the calls.  All ranks must make the same calls with the
 * same lo/hi.  tpc_pass1_insert / tpc_pass1_query refuse to run on a sharded context.
 *
 *   tpc_shard_config   rank/world; reallocates the filter to the 2^L/world-bit shard
 *   tpc_shard_plan     geometry of one pass (TPC_SHARD_INSERT / TPC_SHARD_QUERY), geom[16]:
 *                        [0] batches  [1] tiles (of 16384 positions) per rank and batch
 *                        [2] bytes of one destination block of the region buffer
 *                        [3] bytes of one destination block of the count buffer
 *                        [4] survivor capacity (entries)  [5] overflow capacity (entries)
 *                        [6] bytes per overflow entry  [7] slice_bits  [8] b1  [9] b2
 *                        [10] slice permutation multiplier  [11] its inverse  [12] b3 (0: two levels; filters beyond
 *                        2^38 bits take a third level on the owner: the exchange stays at level 1)
 *                      send and receive buffers hold `world` blocks each
 *   tpc_shard_hash     level 1 of the pass over this rank's tiles of `batch` into send_regions /
 *                      send_counts (block d = entries for rank d); the query also marks the
 *                      N-adjacent vertices of those tiles; *n_overflow = entries that did not fit
 *                      a region (>= 2^62: the overflow list itself overflowed -> unsupported skew)
 *   tpc_shard_overflow_get / _set   the overflow list of the pass (full addresses, any owner):
 *                      ranks all-gather their lists and set the concatenation before _apply
 *   tpc_shard_apply    levels 2-3 over the received blocks: insert ORs the owned slices; query
 *                      tests the first probe of every received edge and keeps the hits as the
 *                      survivor list (*n_survivors; ids relative to the batch)
 *   tpc_shard_pack / tpc_shard_apply_packed   exact-size exchange instead of the equal blocks: the regions have a fixed
 *                      capacity and are ~3/4 full, so after tpc_shard_hash the used prefix of every region is packed
 *                      (block d of packed_dev = the entries for rank d, bytes_per_dest_host[d] bytes, a multiple of 128);
 *                      the host layer moves the count blocks as before (equal all_to_all) and the packed blocks with a
 *                      variable all_to_all, source blocks back to back in rank order, and tpc_shard_apply_packed places
 *                      every received region by a scan of the received counts.  packed_dev holds up to world blocks of
 *                      geom[2] bytes (as the region buffer)
 *   tpc_shard_survivors       copy the survivor ids to a device buffer
 *   tpc_shard_survivor_sources   the rank that hashed each survivor's position (it rides in the id): survivors go BACK to
 *                             that rank (route + variable all_to_all) and are verified there, where their text is -- a
 *                             rank then needs only its own chunk of the packed text (tpc_set_option "text_window")
 *   tpc_shard_verify_addrs    for hash functions fn .. fn+fn_count-1: owner rank and shard-local bit
 *                             address of every survivor id in sid_dev (entry i*fn_count + j)
 *   tpc_shard_probe           answer probes against this rank's shard (hit_dev[i] = 0/1)
 *   tpc_shard_mark            set the candidate mark of every id in sid_dev (all q probes hit)
 *   tpc_shard_route           owner-major send order for `n` probes: perm_dev[i] = slot of item i, counts_host[r] =
 *                             items for rank r (the grouping a host language would do with a sort; world <= 64)
 *   tpc_shard_permute64       dst[perm[i]] = src[i] (the addresses into send order)
 *   tpc_shard_select          the ids of sid_dev whose fn_count answers are all 1; hit_dev holds the answers in SEND
 *                             order (owners answer in the order they were asked), sid_out_dev / *n_out the kept ids
 *   tpc_mask_export / tpc_mask_merge   round mask to / OR of `count` masks from a device buffer:
 *                             the union over ranks is the mask tpc_pass1_query would produce
 *   tpc_mask_export_padded / tpc_mask_or_blocks / tpc_mask_import   the same union as an OR all-reduce by word
 *                             ranges: export padded to world x chunk words, all_to_all of the chunks, fold the
 *                             `count` received chunks, all_gather the folded chunks, import (2 (W-1)/W mask sizes
 *                             per rank on the wire instead of W) */
#define TPC_SHARD_INSERT 0
#define TPC_SHARD_QUERY 1
int tpc_shard_config(tpc_ctx *ctx, uint32_t rank, uint32_t world);
int tpc_shard_plan(tpc_ctx *ctx, int pass, uint64_t lo, uint64_t hi, uint64_t *geom);
int tpc_shard_hash(tpc_ctx *ctx, int pass, uint64_t batch, uint64_t lo, uint64_t hi, void *send_regions_dev, void *send_counts_dev,
                   uint64_t *n_overflow);
/* Overlap of hashing and exchange (round 4).  The level-1 hash of a pass reads the text and writes the caller's send
 * buffers, the pass' produced overflow list and (query) the round mask -- nothing the exchange or the apply of another batch, or of the
 * other pass, touches: the query's hash does not depend on the round's insert at all (reference: CandidateCheckingWorker hashes the same
 * windows FilterFillerWorker did, vertexenumerator.h:633-674 / 1035-1083).
 *   tpc_shard_plan_both   plans insert and query of a round together: shared buffers sized for the larger need, both plans valid
 *   tpc_shard_hash_begin  enqueues the hash of (pass, batch) on the context's second stream and returns at once; the send buffers
 *                         must not be the ones an exchange still reads (double buffering is the caller's); one hash in flight per pass
 *   tpc_shard_hash_end    waits for it; *n_overflow as tpc_shard_hash
 * The overflow lists are two per pass: hashes append to the produced list (tpc_shard_overflow_get), tpc_shard_overflow_set fills the
 * applied one, so a hash running under batch b's exchange cannot disturb the entries batch b's apply is about to consume. */
int tpc_shard_plan_both(tpc_ctx *ctx, uint64_t lo, uint64_t hi, uint64_t *geom_insert /* [16] */, uint64_t *geom_query /* [16] */);
int tpc_shard_hash_begin(tpc_ctx *ctx, int pass, uint64_t batch, uint64_t lo, uint64_t hi, void *send_regions_dev, void *send_counts_dev);
int tpc_shard_hash_end(tpc_ctx *ctx, int pass, uint64_t *n_overflow);
int tpc_shard_overflow_get(tpc_ctx *ctx, int pass, void *dst_dev, uint64_t n);
int tpc_shard_overflow_set(tpc_ctx *ctx, int pass, const void *src_dev, uint64_t n);
int tpc_shard_apply(tpc_ctx *ctx, int pass, uint64_t batch, const void *recv_regions_dev, const void *recv_counts_dev, uint64_t *n_survivors);
int tpc_shard_pack(tpc_ctx *ctx, int pass, const void *send_regions_dev, const void *send_counts_dev, void *packed_dev, uint64_t *bytes_per_dest_host);
int tpc_shard_apply_packed(tpc_ctx *ctx, int pass, uint64_t batch, const void *recv_packed_dev, const void *recv_counts_dev, uint64_t *n_survivors);
/* tpc_shard_apply with this rank's own block read in place: block `rank` of the receive buffers is never touched (the host layer need not
 * copy it), the entries this rank hashed for its own slices are taken from the send buffers tpc_shard_hash filled -- same block index,
 * same layout.  world == 1: the receive buffers may be null.  Option "shard_tight_regions" (default 1) sizes the level-1 regions of the
 * sharded passes at their expected fill + 6 sigma (+ the gate's share in a multi-round pass) instead of the one-GPU 1.3 x: the equal
 * blocks then carry a few per cent of slack and the packing pass (tpc_shard_pack: a read and a write of every entry) can be skipped. */
int tpc_shard_apply_inplace(tpc_ctx *ctx, int pass, uint64_t batch, const void *recv_regions_dev, const void *recv_counts_dev,
                            const void *send_regions_dev, const void *send_counts_dev, uint64_t *n_survivors);
int tpc_shard_survivors(tpc_ctx *ctx, uint64_t *sid_dev);
/* The verification's bookkeeping in fused form (what the calls above do in three or four steps each; same protocol, same results):
 *   tpc_shard_survivors_home  the survivors of the last tpc_shard_apply grouped by the rank that hashed their position, into send_dev
 *                             (n entries, the n tpc_shard_apply returned); counts_host[r] = survivors for rank r (variable all_to_all
 *                             of 8 bytes each); tmp_dev: n uint64 of scratch, unused when world == 1
 *   tpc_shard_verify_send     shard-local bit addresses of hash functions fn .. fn+fn_count-1 of every survivor id in sid_dev, already in
 *                             owner-major send order: send_dev[n * fn_count], perm_dev[i * fn_count + j] = slot of survivor i's j-th
 *                             probe, counts_host[r] = probes for rank r; tmp_dev: n * fn_count uint64 of scratch.  world == 1: the
 *                             natural order is the send order; tmp_dev and perm_dev are not used (pass perm_dev = NULL on)
 *   tpc_shard_finish          last round of a batch: the candidate mark of every survivor whose fn_count answers (hit_dev, in send
 *                             order; perm_dev NULL = natural order) are all 1 -- tpc_shard_select + tpc_shard_mark without the list
 *                             in between; *n_marked = how many passed.  tpc_shard_select takes perm_dev = NULL the same way. */
int tpc_shard_survivors_home(tpc_ctx *ctx, uint64_t *tmp_dev, uint64_t *send_dev, uint64_t *counts_host);
int tpc_shard_verify_send(tpc_ctx *ctx, int fn, int fn_count, const uint64_t *sid_dev, uint64_t n, uint64_t *tmp_dev, uint64_t *send_dev, uint32_t *perm_dev,
                          uint64_t *counts_host);
int tpc_shard_finish(tpc_ctx *ctx, const uint64_t *sid_dev, uint64_t n, int fn_count, const uint8_t *hit_dev, const uint32_t *perm_dev, uint64_t *n_marked);
int tpc_shard_survivor_sources(tpc_ctx *ctx, const uint64_t *sid_dev, uint64_t n, int32_t *source_dev);
int tpc_shard_verify_addrs(tpc_ctx *ctx, int fn, int fn_count, const uint64_t *sid_dev, uint64_t n, uint64_t *addr_dev, int32_t *owner_dev);
int tpc_shard_probe(tpc_ctx *ctx, const uint64_t *addr_dev, uint64_t n, uint8_t *hit_dev);
int tpc_shard_mark(tpc_ctx *ctx, const uint64_t *sid_dev, uint64_t n);
int tpc_shard_route(tpc_ctx *ctx, const int32_t *owner_dev, uint64_t n, uint32_t *perm_dev, uint64_t *counts_host);
int tpc_shard_permute64(tpc_ctx *ctx, const uint64_t *src_dev, const uint32_t *perm_dev, uint64_t n, uint64_t *dst_dev);
int tpc_shard_select(tpc_ctx *ctx, const uint64_t *sid_dev, uint64_t n, int fn_count, const uint8_t *hit_dev, const uint32_t *perm_dev,
                     uint64_t *sid_out_dev, uint64_t *n_out);
int tpc_mask_export(tpc_ctx *ctx, uint32_t *dst_dev);
int tpc_mask_merge(tpc_ctx *ctx, const uint32_t *src_dev, uint32_t count);
int tpc_mask_export_padded(tpc_ctx *ctx, uint32_t *dst_dev, uint64_t total_words);
int tpc_mask_or_blocks(tpc_ctx *ctx, const uint32_t *blocks_dev, uint32_t count, uint64_t words, uint32_t *out_dev);
int tpc_mask_import(tpc_ctx *ctx, const uint32_t *src_dev);

/* ---- combined exchange: the filter REPLICATED through set-bit lists (multi-GPU, round 6) --------------------------------------
 * The reference's threads share one ConcurrentBitVector for free (fetch_or, concurrentbitvector.cpp:31-45; MergeOr :115-122).  Routing
 * every hash hit to the rank that owns its slice (tpc_shard_* above) moves 4 bytes per insert address and 8 per query probe; but an
 * insert matters only the first time a bit is set, and a rank's write-combining passes OR its inserts into LDS slices anyway.  Here
 * every rank keeps the WHOLE filter (option "replicate_filter" = 1 before tpc_shard_config / tpc_set_params; sensible while 2^L / 8
 * bytes fit one GPU beside the partition buffers) and hashes only its chunk of the text:
 *   tpc_filter_reset, tpc_pass1_insert   the one-GPU insert over this rank's chunk of the tiles (tpc_shard_chunk); with one tile
 *                          batch its apply is deferred: the entries wait in their level-2 regions
 *   tpc_combine_info       info[0] = 1: they do, lists can be exported ([1] slices, [2] 2^16-bit windows per slice, [3] upper bound of
 *                          a destination block in 16-byte units for `n_dest` destinations, [4] slice_bits, [5] b1, [6] b2, [7] directory
 *                          entries per destination block); info[0] = 0: the insert was applied to this rank's dense filter (several
 *                          batches, three levels, no memory): OR-reduce the filters instead (tpc_filter_copy_out / tpc_mask_or_blocks /
 *                          tpc_filter_copy_in: an all_to_all of word ranges, a fold, an all_gather)
 *   tpc_combine_export     every slice built in LDS from the rank's own entries; its SET BITS leave as 16-bit offsets per
 *                          2^16-bit window (csrc/tpc_lists.h): block d of payload_dev (cap_units 16-byte units each) = the slices of
 *                          the level-1 buckets b1 % n_dest == d in the order [b1 / n_dest][b2]; dir_dev[d][slice][window] =
 *                          first unit << 24 | entries; units_host[d] = units used.  2 bytes per DISTINCT set bit of the chunk
 *   (exchange, the host's: an all-gather of the exports, blocks and directories; or a reduce-scatter -- block d and its directory to rank d)
 *   tpc_combine_merge      reduce-scatter only: the owner ORs the n_src = world received blocks (source s at unit src_base_host[s], its
 *                          directory at dir_dev + s * stride, stride = info[7]) slice by slice and emits the merged lists of ITS slices
 *                          (one block, directory [slice][window]); the sum of the received units always suffices as capacity
 *   (all-gather of the merged blocks and their directories)
 *   tpc_combine_import     the lists every slice is to be built from: n_src blocks (block s at unit src_base_host[s], its directory at
 *                          dir_dev + s * dir_stride).  n_owner = W > 0 (all-gathered blocks): block s lists only the slices of the
 *                          level-1 buckets b1 % W == s % W, keyed [b1 / W][b2] -- the W merged blocks of tpc_combine_merge (n_src = W),
 *                          or all W x W blocks of the ranks' exports, rank-major (n_src = W * W, no reduce-scatter); n_owner = 0: every
 *                          block lists every slice (exports with n_dest = 1).  The insert is then pending again: the buffers must stay
 *                          untouched until the query ran
 *   tpc_pass1_query        the one-GPU query over this rank's chunk; its first lookup builds every slice from the lists, writes it to
 *                          the rank's filter and tests the chunk's probes in LDS: no probe, no survivor, no answer crosses a link.
 *                          The round mask then holds the marks of this rank's positions (as after tpc_shard_finish: second pass sharded
 *                          by key hash, tpc_pass2_marks ...)
 *   tpc_combine_choose     the bytes model both hosts print and follow: bytes a rank receives per round for (1) all-gather of exports,
 *                          (2) reduce-scatter + all-gather of merged lists, (3) dense OR all-reduce, from the mean export size; returns
 *                          the cheapest.  No context: pure arithmetic. */
int tpc_combine_info(tpc_ctx *ctx, uint32_t n_dest, uint64_t *info /* [8] */);
int tpc_combine_export(tpc_ctx *ctx, uint32_t n_dest, uint16_t *payload_dev, uint64_t cap_units, uint64_t *dir_dev, uint64_t *units_host);
int tpc_combine_merge(tpc_ctx *ctx, uint32_t n_src, const uint16_t *payload_dev, const uint64_t *src_base_host, const uint64_t *dir_dev, uint16_t *out_payload_dev,
                      uint64_t out_cap_units, uint64_t *out_dir_dev, uint64_t *units_host);
int tpc_combine_import(tpc_ctx *ctx, uint32_t n_src, uint32_t n_owner, const uint16_t *payload_dev, const uint64_t *src_base_host, const uint64_t *dir_dev, uint64_t dir_stride);
int tpc_combine_choose(uint32_t world, int L, uint64_t mean_export_units, double *bytes /* [3] */);
/* Words [word0, word0 + n_words) of the filter to / from a device buffer (the dense form of the exchange; a pending insert is applied first). */
int tpc_filter_copy_out(tpc_ctx *ctx, uint64_t word0, uint64_t n_words, uint32_t *dst_dev);
int tpc_filter_copy_in(tpc_ctx *ctx, uint64_t word0, uint64_t n_words, const uint32_t *src_dev);

/* ---- parity taps (debug; used by tests/) ---------------------------------------------- */
uint64_t tpc_filter_words(const tpc_ctx *ctx);               /* 2^L/32 + 1, concurrentbitvector.cpp:12 (sharded: 2^L/32/world) */
int tpc_filter_download(tpc_ctx *ctx, uint32_t *words_host); /* tpc_filter_words words       */
/* Restore a downloaded filter (the reference's commented-out ReloadBloomFilter, VE.h:29,113-121): a
 * checkpoint of the most expensive state of a run; the next tpc_pass1_query uses these bits. */
int tpc_filter_upload(tpc_ctx *ctx, const uint32_t *words_host);
uint64_t tpc_mask_words(const tpc_ctx *ctx);                 /* n_text/32 + 1                 */
int tpc_mask_download(tpc_ctx *ctx, int run_wide, uint32_t *words_host);
/* Vertex hashes of the windows at g0..g0+n-1: out[(g-g0)*2q + 2i] = pos_i, +1 = neg_i. */
int tpc_hash_dump(tpc_ctx *ctx, uint64_t g0, uint64_t n, uint64_t *out_host);

/* ---- measurement ------------------------------------------------------------------------ */
/* Duration in ms of the most recent launch(es) of kernel `which`, measured with hipEvents on
 * the stream the kernel ran on; <0 if it has not run. */
double tpc_kernel_ms(const tpc_ctx *ctx, int which);
/* Tuning knobs (results never depend on them):
 *   insert_test_first  direct insert kernel: 0 = atomicOr per address, 1 = test-then-atomicOr (VE.h:1088)
 *   insert_mode / query_mode   0 = automatic, 1 = direct scattered kernel, 2 = LDS write-combining passes
 *   slice_bits         log2 bits of a filter slice held in LDS (6..20, default 20)
 *   part_levels        0 = automatic (three binning levels when L - slice_bits > 18), 2, 3
 *   text_window        1: on a sharded context tpc_seq_upload keeps only the words of the tiles this rank hashes (chunk
 *                      rank * ceil(tiles / world) ..., + halo); such a context runs the sharded first pass only (rank 0 of the
 *                      C++ host keeps the whole text for the second pass)
 *   fuse_apply_lookup  1 (default): when the insert of a round fits one tile batch and the query is partitioned with the same
 *                      geometry, the insert stops after its level-2 binning and the lookup kernel of the query's FIRST batch builds
 *                      each filter slice itself (the filter is written once and not read back by that batch); TPC_K_INSERT then
 *                      covers hash + split only and TPC_K_FUSED the shared kernel; 0: off
 *   test_sched_cap     tests only, process-wide: rounds per segment of the split kernels' round schedule (0 = what fits in LDS)
 *   test_fail_mallocs  tests only, process-wide: the next N second-pass / output allocations fail at their first attempt, as if
 *                      the device were full (they then give the partition buffers back and try again, see "pbuf_releases")
 *   test_force_anyq    tests only, process-wide: 1 = the closed-form first-pass kernels that serve q = 17..64
 *                      (csrc/tpc_pass1_anyq.hip) for every q, so that they can be checked on the goldens with q <= 16
 *   part_budget_bytes  partition buffers per tile batch (0 = automatic: 40 GiB, or 60 % of the free device
 *                      memory when that is more; any number of batches, not only powers of two); part_min_tiles  smallest batch */
/*   replicate_filter   1 (before tpc_shard_config / tpc_set_params): a sharded context keeps the whole filter; tpc_pass1_insert / tpc_pass1_query
 *                      run over this rank's chunk of the tiles and the tpc_combine_* calls exchange set-bit lists (see there) */
int tpc_set_option(tpc_ctx *ctx, const char *name, int64_t value);
/* What the last first-pass calls ran: "insert_path" / "query_path" = 1 direct kernel, 2 or 3 = LDS
 * write-combining with that many levels (+10: it overflowed and the direct kernel completed the pass);
 * "insert_batches" / "query_batches" = tile batches; "filter2_retries" = exact-filter passes repeated
 * with the full-size table by the last tpc_pass2_filter; "text_words" = packed words of the text held (a window with option text_window); "fused_lookups" = queries that built the filter slices themselves (deferred apply); "query_overflow_entries" = entries the last batch of the last partitioned query handed to its overflow list (full rings or regions: address skew); "pbuf_releases" = times a second-pass or output allocation did not fit beside the first pass' partition buffers, which were then freed (the next first pass allocates them again); "round_marks" = candidate marks of the round the last
 * tpc_pass2_filter consumed (what tpc_pass1_query reports; the sharded first pass has no single call that does);
 * "device_free_bytes" / "device_total_bytes" = hipMemGetInfo of the context's device, now.
 * -1: unknown name. */
int64_t tpc_get_stat(const tpc_ctx *ctx, const char *name);

#ifdef __cplusplus
}
#endif
#endif /* TWOPACO_HIP_H_ */
