// junctiondump.cpp -- the `graphdump` tool: converts de_bruijn.bin (junctionapi.h) to text formats.
//
// Downstream consumer of the junction stream, kept flag- and byte-compatible with the reference's
// graphdump (reference src/graphdump/graphdump.cpp) so that pipelines built on it are unchanged:
//   graphdump <infile> -f seq|group|dot|gfa1|gfa2|fasta -k <k> [-s <fasta>]... [--prefix]
// Formats (reference line numbers):
//   seq    "chr pos id" per junction occurrence, file order (:160-168)
//   group  occurrences of the same junction id on one line, lines ordered by their first position (:122-158)
//   dot    two arcs per pair of consecutive junctions of a sequence (:588-609)
//   gfa1 / gfa2 / fasta  the compacted graph: one segment per pair of consecutive junctions; its id packs the
//          smaller-id end junction, its strand and the following character (:44-113), or a fresh id from
//          2^34 upwards when that character is 'N'; segments are printed on first sight, then the occurrence
//          (C / F line), the link to the previous segment of the sequence (L / E) and one path per sequence
//          (P / O) (:379-480, :504-585).
// Differences from the reference that do not change the output: the "seen" set is a hash set instead of a
// 2^35-bit vector (4 GiB), and output is buffered.  Where the reference reads out of bounds (a .bin whose first
// sequences were too short to be dispatched, so that sequence ids and FASTA records get out of step) it prints
// garbage before its "The input is corrupted"; this tool reports the error without the garbage.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

#include "dnachar.h"
#include "junctionapi.h"
#include "streamfastaparser.h"

namespace
{
	using TwoPaCo::DnaChar;
	using TwoPaCo::JunctionPosition;

	// ---------------------------------------------------------------------------------------- output
	class Out
	{
	public:
		~Out() { Flush(); }
		Out & operator << (const std::string & s) { buf_ += s; return Check(); }
		Out & operator << (const char * s) { buf_ += s; return Check(); }
		Out & operator << (char c) { buf_ += c; return Check(); }
		Out & operator << (int64_t v) { buf_ += std::to_string(static_cast<long long>(v)); return Check(); }
		Out & operator << (uint64_t v) { buf_ += std::to_string(static_cast<unsigned long long>(v)); return Check(); }
		Out & operator << (uint32_t v) { buf_ += std::to_string(v); return Check(); }
		void Flush()
		{
			if (!buf_.empty()) std::fwrite(buf_.data(), 1, buf_.size(), stdout);
			buf_.clear();
			std::fflush(stdout);
		}

	private:
		Out & Check()
		{
			if (buf_.size() > (1u << 20))
			{
				std::fwrite(buf_.data(), 1, buf_.size(), stdout);
				buf_.clear();
			}

			return *this;
		}

		std::string buf_;
	};

	int64_t Magnitude(int64_t x) { return x < 0 ? -x : x; }
	char Strand(int64_t x) { return x >= 0 ? '+' : '-'; }

	// ---------------------------------------------------------------------------------------- seq / group / dot
	void DumpSeq(const std::string & binFile, Out & out)
	{
		TwoPaCo::JunctionPositionReader reader(binFile);
		for (JunctionPosition j; reader.NextJunctionPosition(j);)
		{
			out << uint64_t(j.GetChr()) << ' ' << uint64_t(j.GetPos()) << ' ' << int64_t(j.GetId()) << '\n';
		}
	}

	void DumpGroups(const std::string & binFile, Out & out)
	{
		struct Occ { int64_t id; uint32_t chr, pos; };
		std::vector<Occ> all;
		TwoPaCo::JunctionPositionReader reader(binFile);
		for (JunctionPosition j; reader.NextJunctionPosition(j);)
		{
			all.push_back(Occ{j.GetId(), j.GetChr(), j.GetPos()});
		}

		// one group per signed id, members in (chr, pos) order; groups in the order of their first member
		std::sort(all.begin(), all.end(), [](const Occ & a, const Occ & b)
		{
			if (a.id != b.id) return a.id < b.id;
			return a.chr != b.chr ? a.chr < b.chr : a.pos < b.pos;
		});

		std::vector<std::pair<size_t, size_t> > group;
		for (size_t i = 0; i < all.size();)
		{
			size_t j = i;
			while (j < all.size() && all[j].id == all[i].id) ++j;
			group.push_back(std::make_pair(i, j));
			i = j;
		}

		std::stable_sort(group.begin(), group.end(), [&all](const std::pair<size_t, size_t> & a, const std::pair<size_t, size_t> & b)
		{
			const Occ & x = all[a.first];
			const Occ & y = all[b.first];
			return x.chr != y.chr ? x.chr < y.chr : x.pos < y.pos;
		});

		for (const std::pair<size_t, size_t> & g : group)
		{
			for (size_t i = g.first; i < g.second; i++) out << uint64_t(all[i].chr) << ' ' << uint64_t(all[i].pos) << "; ";
			out << '\n';
		}
	}

	void DumpDot(const std::string & binFile, Out & out)
	{
		out << "digraph G\n{\n\trankdir = LR\n";
		TwoPaCo::JunctionPositionReader reader(binFile);
		JunctionPosition prev;
		for (JunctionPosition now; reader.NextJunctionPosition(now); prev = now)
		{
			if (now.GetChr() != prev.GetChr()) continue;
			out << '\t' << int64_t(prev.GetId()) << " -> " << int64_t(now.GetId()) << "[color=\"blue\", label=\"chr=" << uint64_t(prev.GetChr()) << " pos="
				<< uint64_t(prev.GetPos()) << "\"]\n";
			out << '\t' << int64_t(-now.GetId()) << " -> " << int64_t(-prev.GetId()) << "[color=\"red\", label=\"chr=" << uint64_t(prev.GetChr()) << " pos="
				<< uint64_t(prev.GetPos()) << "\"]\n";
		}

		out << "}\n";
	}

	// ---------------------------------------------------------------------------------------- segments
	// Segment naming, reference graphdump.cpp:44-113.  The segment between junctions a (left) and b (right) is
	// oriented from its end with the smaller |id| (ties: forward unless both ids are 0); `next` is the character
	// that follows the start junction's k-mer in that orientation.
	class SegmentNamer
	{
	public:
		SegmentNamer() : nextUnique_(int64_t(1) << 34) {}

		int64_t Name(int64_t leftId, int64_t rightId, char afterLeft, char beforeRightComplemented)
		{
			const int64_t LIMIT = int64_t(1) << 31;  // MAX_JUNCTION_ID
			const int64_t l = Magnitude(leftId), r = Magnitude(rightId);
			if (l >= LIMIT || r >= LIMIT)
			{
				throw std::runtime_error("A vertex id is too large, cannot generate GFA");
			}

			const bool forward = l < r || (l == r && l > 0);
			const char next = forward ? afterLeft : beforeRightComplemented;
			const int64_t start = forward ? leftId : -rightId;
			if (next == 'N')
			{
				return nextUnique_++;
			}

			int64_t name = static_cast<int64_t>(DnaChar::MakeUpChar(next));
			if (start < 0)
			{
				name |= 1 << 2;
				name |= Magnitude(start) << 3;
			}
			else
			{
				name |= start << 3;
			}

			return forward ? name : -name;
		}

	private:
		int64_t nextUnique_;
	};

	struct InputSequences
	{
		std::vector<std::string> name;
		std::vector<uint64_t> length;
		std::map<std::string, std::string> file;
	};

	void ListSequences(const std::vector<std::string> & fasta, bool prefixed, InputSequences & seq)
	{
		size_t index = 0;  // the reference never advances this counter: every prefix is "s0_" (graphdump.cpp:176-192)
		for (const std::string & f : fasta)
		{
			TwoPaCo::StreamFastaParser parser(f);
			while (parser.ReadRecord())
			{
				const std::string id = prefixed ? "s" + std::to_string(index) + "_" + parser.GetCurrentHeader() : parser.GetCurrentHeader();
				seq.name.push_back(id);
				seq.file[id] = f;
				uint64_t n = 0;
				for (char ch; parser.GetChar(ch);) ++n;
				seq.length.push_back(n);
			}
		}
	}

	// one sequence after the other, across the files
	class SequenceCursor
	{
	public:
		explicit SequenceCursor(const std::vector<std::string> & fasta) : fasta_(fasta), file_(0), parser_(0)
		{
			if (!fasta_.empty()) parser_ = new TwoPaCo::StreamFastaParser(fasta_[0]);
		}

		~SequenceCursor() { delete parser_; }

		bool Next(std::string & body)
		{
			body.clear();
			while (file_ < fasta_.size())
			{
				if (parser_->ReadRecord())
				{
					for (char ch; parser_->GetChar(ch);) body.push_back(ch);
					return true;
				}

				delete parser_;
				parser_ = 0;
				if (++file_ < fasta_.size()) parser_ = new TwoPaCo::StreamFastaParser(fasta_[file_]);
			}

			return false;
		}

	private:
		SequenceCursor(const SequenceCursor &);
		void operator = (const SequenceCursor &);
		std::vector<std::string> fasta_;
		size_t file_;
		TwoPaCo::StreamFastaParser * parser_;
	};

	struct SegmentEvent
	{
		int64_t id;            // signed name
		uint64_t size;
		bool first;            // first sight of |id|
		uint64_t begin, end;   // junction positions in the sequence
		size_t sequence;
	};

	class SegmentSink
	{
	public:
		virtual ~SegmentSink() {}
		virtual void Segment(const SegmentEvent & e, const std::string & chr, size_t k) = 0;
		virtual void EndOfSequence(size_t sequence) = 0;
	};

	// The walk shared by gfa1 / gfa2 / fasta (reference graphdump.cpp:398-480).  Consecutive records of the same
	// sequence bound a segment; a change of sequence must step the sequence id by exactly one.
	void WalkSegments(const std::string & binFile, const std::vector<std::string> & fasta, size_t k, SegmentSink & sink)
	{
		SegmentNamer namer;
		std::unordered_set<int64_t> seen;
		SequenceCursor cursor(fasta);
		TwoPaCo::JunctionPositionReader reader(binFile);
		std::string chr;
		size_t sequence = 0;
		JunctionPosition left;
		if (!reader.NextJunctionPosition(left))
		{
			sink.EndOfSequence(sequence);
			return;
		}

		// The reference indexes sequence 0 with whatever the first record holds (out of bounds when the first
		// sequences are shorter than k and were skipped); here that input is reported as what it is.
		cursor.Next(chr);
		if (left.GetChr() != 0)
		{
			throw std::runtime_error("The input is corrupted");
		}

		for (JunctionPosition right; reader.NextJunctionPosition(right); left = right)
		{
			if (left.GetChr() != right.GetChr())
			{
				sink.EndOfSequence(sequence);
				cursor.Next(chr);
				if (right.GetChr() != ++sequence)
				{
					throw std::runtime_error("The input is corrupted");
				}

				continue;
			}

			if (right.GetPos() <= left.GetPos() || uint64_t(right.GetPos()) + k > chr.size())
			{
				throw std::runtime_error("The input is corrupted");
			}

			SegmentEvent e;
			e.id = namer.Name(left.GetId(), right.GetId(), chr[left.GetPos() + k], DnaChar::ReverseChar(chr[right.GetPos() - 1]));
			e.size = uint64_t(right.GetPos()) + k - left.GetPos();
			e.first = seen.insert(Magnitude(e.id)).second;
			e.begin = left.GetPos();
			e.end = right.GetPos();
			e.sequence = sequence;
			sink.Segment(e, chr, k);
		}

		sink.EndOfSequence(sequence);
	}

	std::string SegmentBody(const SegmentEvent & e, const std::string & chr, size_t k)
	{
		const std::string body = chr.substr(e.begin, e.end + k - e.begin);
		return e.id > 0 ? body : DnaChar::ReverseCompliment(body);
	}

	// ---------------------------------------------------------------------------------------- GFA
	class GfaSink : public SegmentSink
	{
	public:
		GfaSink(Out & out, const InputSequences & seq) : out_(out), seq_(seq), prevId_(0), prevSize_(0) {}

		void Segment(const SegmentEvent & e, const std::string & chr, size_t k)
		{
			if (e.first) SegmentLine(e, SegmentBody(e, chr, k));
			Occurrence(e, k);
			if (prevId_ != 0) Link(prevId_, prevSize_, e.id, e.size, k);
			prevId_ = e.id;
			prevSize_ = e.size;
			path_.push_back(e.id);
		}

		void EndOfSequence(size_t sequence)
		{
			if (!path_.empty()) Path(seq_.name[sequence]);
			path_.clear();
			prevId_ = 0;
		}

	protected:
		virtual void SegmentLine(const SegmentEvent & e, const std::string & body) = 0;
		virtual void Occurrence(const SegmentEvent & e, size_t k) = 0;
		virtual void Link(int64_t a, uint64_t aSize, int64_t b, uint64_t bSize, size_t k) = 0;
		virtual void Path(const std::string & name) = 0;
		Out & out_;
		const InputSequences & seq_;
		std::vector<int64_t> path_;

	private:
		int64_t prevId_;
		uint64_t prevSize_;
	};

	class Gfa1Sink : public GfaSink
	{
	public:
		Gfa1Sink(Out & out, const InputSequences & seq) : GfaSink(out, seq) {}

	protected:
		void SegmentLine(const SegmentEvent & e, const std::string & body) { out_ << "S\t" << Magnitude(e.id) << '\t' << body << '\n'; }

		void Occurrence(const SegmentEvent & e, size_t)
		{
			out_ << "C\t" << Magnitude(e.id) << '\t' << Strand(e.id) << '\t' << seq_.name[e.sequence] << "\t+\t" << e.end << '\n';
		}

		void Link(int64_t a, uint64_t, int64_t b, uint64_t, size_t k)
		{
			out_ << "L\t" << Magnitude(a) << '\t' << Strand(a) << '\t' << Magnitude(b) << '\t' << Strand(b) << '\t' << uint64_t(k) << "M\n";
		}

		void Path(const std::string & name)
		{
			out_ << "P\t" << name << '\t';
			for (size_t i = 0; i < path_.size(); i++) out_ << Magnitude(path_[i]) << Strand(path_[i]) << (i + 1 < path_.size() ? "," : "\t*\n");
		}
	};

	class Gfa2Sink : public GfaSink
	{
	public:
		Gfa2Sink(Out & out, const InputSequences & seq) : GfaSink(out, seq) {}

	protected:
		static std::string At(uint64_t pos, uint64_t length) { return pos == length ? std::to_string(pos) + "$" : std::to_string(pos); }

		void SegmentLine(const SegmentEvent & e, const std::string & body) { out_ << "S\t" << Magnitude(e.id) << '\t' << e.size << '\t' << body << '\n'; }

		void Occurrence(const SegmentEvent & e, size_t k)
		{
			const uint64_t total = seq_.length[e.sequence];
			out_ << "F\t" << Magnitude(e.id) << '\t' << seq_.name[e.sequence] << Strand(e.id) << "\t0\t" << e.size << "$\t" << At(e.begin, total) << '\t'
				<< At(e.end + k, total) << '\t' << uint64_t(k) << "M\n";
		}

		void Link(int64_t a, uint64_t aSize, int64_t b, uint64_t bSize, size_t k)
		{
			const uint64_t a0 = a > 0 ? aSize - k : 0, a1 = a > 0 ? aSize : k;   // the overlapping k-mer on each segment
			const uint64_t b0 = b > 0 ? 0 : bSize - k, b1 = b > 0 ? k : bSize;
			out_ << "E\t" << Magnitude(a) << Strand(a) << '\t' << Magnitude(b) << Strand(b) << '\t' << At(a0, aSize) << '\t' << At(a1, aSize) << '\t'
				<< At(b0, bSize) << '\t' << At(b1, bSize) << '\t' << uint64_t(k) << "M\n";
		}

		void Path(const std::string & name)
		{
			out_ << "O\t" << name << "p\t";
			for (size_t i = 0; i < path_.size(); i++) out_ << Magnitude(path_[i]) << Strand(path_[i]) << (i + 1 < path_.size() ? " " : "\n");
		}
	};

	class FastaSink : public SegmentSink
	{
	public:
		explicit FastaSink(Out & out) : out_(out) {}

		void Segment(const SegmentEvent & e, const std::string & chr, size_t k)
		{
			if (!e.first) return;
			out_ << '>' << Magnitude(e.id) << '\n';
			const std::string body = SegmentBody(e, chr, k);
			for (size_t i = 0; i < body.size(); i += 80) out_ << body.substr(i, 80) << '\n';
		}

		void EndOfSequence(size_t) {}

	private:
		Out & out_;
	};

	// ---------------------------------------------------------------------------------------- command line
	struct ArgError
	{
		std::string what, arg;
		ArgError(const std::string & w, const std::string & a) : what(w), arg(a) {}
	};

	void Usage()
	{
		std::printf("\nUSAGE: \n\n   graphdump  [-k <integer>] [-s <string>] ... -f <seq|group|dot|gfa1|gfa2|fasta> [--prefix] [--] [--version] [-h] <file name>\n\n"
			"Where: \n\n"
			"   -k <integer>,  --kvalue <integer>\n     (required)  Value of k\n\n"
			"   -s <string>,  --seqfile <string>  (accepted multiple times)\n     sequences file name\n\n"
			"   -f <seq|group|dot|gfa1|gfa2|fasta>,  --format <seq|group|dot|gfa1|gfa2|fasta>\n     (required)  Output format\n\n"
			"   --prefix\n     Add a prefix to segments in GFA (in case if you have genomes with identical FASTA headers)\n\n"
			"   <file name>\n     (required)  input file name\n\n"
			"   This utility converts the binary output of TwoPaCo to another format\n\n");
	}
}

int main(int argc, char * argv[])
{
	try
	{
		std::string binFile, format;
		std::vector<std::string> fasta;
		bool prefix = false, haveK = false, haveFormat = false, haveFile = false;
		size_t k = 25;
		bool positionalOnly = false;
		for (int i = 1; i < argc; i++)
		{
			const std::string a = argv[i];
			auto value = [&](const char * id) -> std::string
			{
				if (i + 1 >= argc) throw ArgError("Missing a value for this argument!", id);
				return argv[++i];
			};

			if (positionalOnly || a.empty() || a[0] != '-')
			{
				if (haveFile) throw ArgError("Argument already set!", "(infile)");
				binFile = a;
				haveFile = true;
			}
			else if (a == "--") positionalOnly = true;
			else if (a == "-h" || a == "--help") { Usage(); return 0; }
			else if (a == "--version") { std::printf("\n%s  version: 0.9.4\n\n", argv[0]); return 0; }
			else if (a == "--prefix") prefix = true;
			else if (a == "-k" || a == "--kvalue")
			{
				const std::string v = value("(--kvalue)");
				char * end = 0;
				const long long parsed = std::strtoll(v.c_str(), &end, 10);
				if (end == v.c_str() || *end != 0 || parsed < 0) throw ArgError("Couldn't read argument value from string '" + v + "'", "(--kvalue)");
				k = size_t(parsed);
				haveK = true;
			}
			else if (a == "-s" || a == "--seqfile") fasta.push_back(value("(--seqfile)"));
			else if (a == "-f" || a == "--format")
			{
				format = value("(--format)");
				static const char * allowed[] = {"seq", "group", "dot", "gfa1", "gfa2", "fasta"};
				if (std::find(allowed, allowed + 6, format) == allowed + 6) throw ArgError("Value '" + format + "' does not meet constraint: seq|group|dot|gfa1|gfa2|fasta", "Argument: -f (--format)");
				haveFormat = true;
			}
			else throw ArgError("Couldn't find match for argument", "(" + a + ")");
		}

		if (!haveK) throw ArgError("Required argument missing: kvalue", " ");
		if (!haveFormat) throw ArgError("Required argument missing: format", " ");
		if (!haveFile) throw ArgError("Required argument missing: infile", " ");
		const bool needsSequences = format == "gfa1" || format == "gfa2" || format == "fasta";
		if (needsSequences && fasta.empty()) throw ArgError("Required argument missing\n", "Argument: seqfilename");

		Out out;
		if (format == "seq") DumpSeq(binFile, out);
		else if (format == "group") DumpGroups(binFile, out);
		else if (format == "dot") DumpDot(binFile, out);
		else
		{
			InputSequences seq;
			if (format == "gfa1")
			{
				out << "H\tVN:Z:1.0\n";
				ListSequences(fasta, prefix, seq);
				for (const std::string & name : seq.name) out << "S\t" << name << "\t*\tUR:Z:" << seq.file[name] << '\n';
				Gfa1Sink sink(out, seq);
				WalkSegments(binFile, fasta, k, sink);
			}
			else if (format == "gfa2")
			{
				out << "H\tVN:Z:2.0\n";
				ListSequences(fasta, prefix, seq);
				Gfa2Sink sink(out, seq);
				WalkSegments(binFile, fasta, k, sink);
			}
			else
			{
				ListSequences(fasta, true, seq);
				FastaSink sink(out);
				WalkSegments(binFile, fasta, k, sink);
			}
		}
	}
	catch (ArgError & e)
	{
		std::fflush(stdout);
		if (e.arg == "Argument: seqfilename")
		{
			std::fprintf(stderr, "error: %s for arg %s\n", e.what.c_str(), e.arg.c_str());  // thrown after parsing, graphdump.cpp:669-693
		}
		else
		{
			std::fprintf(stderr, "PARSE ERROR: %s\n             %s\n\nBrief USAGE: \n   %s  [-k <integer>] [-s <string>] ... -f <seq|group|dot|gfa1|gfa2|fasta> [--prefix] [--] [--version] [-h] <file name>\n\n"
				"For complete USAGE and HELP type: \n   %s --help\n\n", e.arg.c_str(), e.what.c_str(), argv[0], argv[0]);
		}

		return 1;
	}
	catch (std::runtime_error & e)
	{
		std::fflush(stdout);
		std::fprintf(stderr, "error: %s\n", e.what());
		return 1;
	}

	return 0;
}
