// junctionapi.h -- reader/writer of the junction position stream (de_bruijn.bin).
//
// API- and byte-compatible with the reference's header-only junction API
// (reference src/common/junctionapi.h:12-137): downstream tools (graphdump and others) that
// include the reference header can include this one instead.  Format: a flat sequence of
// 12-byte little-endian records {uint32 pos; int64 id}; the sequence (chromosome) index is
// implicit -- a separator record {0xFFFFFFFF, INT64_MAX} advances it by one
// (reference junctionapi.h:81-98 reader, :118-126 writer).  No header, no trailer.
#ifndef _JUNCTION_POSITION_API_H_
#define _JUNCTION_POSITION_API_H_

#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace TwoPaCo
{
	struct JunctionPosition
	{
	public:
		JunctionPosition() : chr_(UINT32_MAX), pos_(UINT32_MAX), bifId_(INT64_MAX) {}
		JunctionPosition(uint32_t chr, uint32_t pos, int64_t bifId) : chr_(chr), pos_(pos), bifId_(bifId) {}
		uint32_t GetPos() const { return pos_; }
		uint32_t GetChr() const { return chr_; }
		int64_t GetId() const { return bifId_; }

	private:
		uint32_t chr_;
		uint32_t pos_;
		int64_t bifId_;
		static const int64_t SEPARATOR_BIF = INT64_MAX;
		static const uint32_t SEPARATOR_POS = UINT32_MAX;
		friend class JunctionPositionReader;
		friend class JunctionPositionWriter;
	};

	class JunctionPositionReader
	{
	public:
		JunctionPositionReader(const std::string & inFileName) : nowChr_(0), in_(inFileName.c_str(), std::ios::binary)
		{
			if (!in_)
			{
				throw std::runtime_error("Can't read the input file");
			}
		}

		// Marks the junction positions of sequence `chr`; stops (and un-reads) at the first record of a later one.
		void RestoreVector(std::vector<bool> & mark, size_t chr)
		{
			JunctionPosition pos;
			mark.assign(mark.size(), false);
			while (NextJunctionPosition(pos))
			{
				if (pos.GetChr() != chr)
				{
					in_.seekg(-static_cast<std::streamoff>(RECORD_BYTES), std::ios::cur);
					break;
				}

				mark[pos.GetPos()] = true;
			}
		}

		void RestoreAllVectors(std::vector<std::vector<bool> > & mark)
		{
			JunctionPosition pos;
			while (NextJunctionPosition(pos))
			{
				mark[pos.GetChr()][pos.GetPos()] = true;
			}
		}

		// Next real record; separators only advance the current sequence index.
		bool NextJunctionPosition(JunctionPosition & pos)
		{
			for (;;)
			{
				uint32_t p = 0;
				int64_t id = 0;
				in_.read(reinterpret_cast<char*>(&p), sizeof(p));
				in_.read(reinterpret_cast<char*>(&id), sizeof(id));
				if (!in_)
				{
					pos = JunctionPosition(nowChr_, p, id);
					return false;
				}

				if (p != JunctionPosition::SEPARATOR_POS && id != JunctionPosition::SEPARATOR_BIF)
				{
					pos = JunctionPosition(nowChr_, p, id);
					return true;
				}

				++nowChr_;
			}
		}

	private:
		static const size_t RECORD_BYTES = sizeof(uint32_t) + sizeof(int64_t);
		uint32_t nowChr_;
		std::ifstream in_;
	};

	class JunctionPositionWriter
	{
	public:
		JunctionPositionWriter(const std::string & outFileName) : nowChr_(0), out_(outFileName.c_str(), std::ios::binary)
		{
			if (!out_)
			{
				throw std::runtime_error("Can't create the output file");
			}

			buffer_.reserve(BUFFER_BYTES + RECORD_BYTES);
		}

		~JunctionPositionWriter()
		{
			Flush();
		}

		void WriteJunction(JunctionPosition pos)
		{
			while (nowChr_ < pos.chr_)
			{
				Put(JunctionPosition::SEPARATOR_POS, JunctionPosition::SEPARATOR_BIF);
				++nowChr_;
			}

			Put(pos.pos_, pos.bifId_);
			if (!out_)
			{
				throw std::runtime_error("Can't write to the output file");
			}
		}

		// Records are staged in a 1 MiB buffer (the reference issues two ofstream writes per record).
		void Flush()
		{
			if (!buffer_.empty())
			{
				out_.write(buffer_.data(), std::streamsize(buffer_.size()));
				buffer_.clear();
			}

			out_.flush();
		}

	private:
		static const size_t RECORD_BYTES = sizeof(uint32_t) + sizeof(int64_t);
		static const size_t BUFFER_BYTES = 1 << 20;
		void Put(uint32_t p, int64_t id)
		{
			const char * a = reinterpret_cast<const char*>(&p);
			const char * b = reinterpret_cast<const char*>(&id);
			buffer_.insert(buffer_.end(), a, a + sizeof(p));
			buffer_.insert(buffer_.end(), b, b + sizeof(id));
			if (buffer_.size() >= BUFFER_BYTES)
			{
				out_.write(buffer_.data(), std::streamsize(buffer_.size()));
				buffer_.clear();
			}
		}

		uint32_t nowChr_;
		std::ofstream out_;
		std::vector<char> buffer_;
	};
}

#endif
