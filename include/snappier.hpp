// snappier.hpp -- header-only C++ mirror of Snappier's static block API (Snappier/Snappy.cs:10-283) over the C-ABI in
// snappier_hip.h.  Same names and error behaviour as the reference: Try* return false for a too-small output buffer,
// InvalidData for corrupt input (InvalidDataException), InvalidOperation for overlapping spans / device failure.
// Every call runs on the GPU; there is no CPU fallback.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "snappier_hip.h"

namespace Snappier {

struct InvalidDataException : std::runtime_error {
    int status;
    explicit InvalidDataException(int st) : std::runtime_error(snp_status_string(st)), status(st) {}
};
struct ArgumentException : std::invalid_argument { using std::invalid_argument::invalid_argument; };
struct InvalidOperationException : std::logic_error { using std::logic_error::logic_error; };

class Context {
public:
    explicit Context(int device = 0, snp_hash_variant hash = SNP_HASH_CRC32C)
    {
        if (snp_ctx_create(device, hash, nullptr, &ctx_) != SNP_OK)
            throw InvalidOperationException("snp_ctx_create failed: no usable HIP device (the codec has no CPU fallback)");
    }
    ~Context() { snp_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    snp_ctx* get() const { return ctx_; }

private:
    snp_ctx* ctx_ = nullptr;
};

inline void ThrowFor(int st, const Context& c)
{
    switch (st) {
        case SNP_OK: return;
        case SNP_ERR_OUTPUT_TOO_SMALL: throw ArgumentException("Output buffer is too small.");   // ThrowHelper.cs:18-19
        case SNP_ERR_BAD_OFFSET: case SNP_ERR_TOO_LONG: case SNP_ERR_INCOMPLETE: case SNP_ERR_BAD_LENGTH:
        case SNP_ERR_CRC_MISMATCH: case SNP_ERR_CHUNK_TYPE: case SNP_ERR_TRUNCATED_STREAM:
            throw InvalidDataException(st);
        case SNP_ERR_BAD_ARG: throw ArgumentException(snp_status_string(st));
        default: throw InvalidOperationException(std::string(snp_status_string(st)) + ": " + snp_ctx_last_error(c.get()));
    }
}

struct Snappy {
    static int GetMaxCompressedLength(int inputLength)                                            // Snappy.cs:20-24
    {
        const int64_t v = snp_max_compressed_length(inputLength);
        if (v < 0) throw ArgumentException("inputLength");
        return static_cast<int>(v);
    }
    static bool TryCompress(Context& c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t& written)   // :55-67
    {
        const int st = snp_try_compress(c.get(), in, n, out, cap, &written);
        if (st == SNP_ERR_OUTPUT_TOO_SMALL) { written = 0; return false; }
        ThrowFor(st, c);
        return true;
    }
    static std::vector<uint8_t> CompressToArray(Context& c, const uint8_t* in, size_t n)           // :99-129
    {
        std::vector<uint8_t> buf(static_cast<size_t>(GetMaxCompressedLength(static_cast<int>(n))));
        size_t w = 0;
        if (!TryCompress(c, in, n, buf.data(), buf.size(), w)) throw InvalidOperationException("unreachable");
        buf.resize(w);
        return buf;
    }
    static int GetUncompressedLength(const uint8_t* in, size_t n)                                 // :136-137
    {
        uint32_t len = 0;
        const int st = snp_get_uncompressed_length(in, n, &len, nullptr);
        if (st != SNP_OK) throw InvalidDataException(st);
        return static_cast<int>(len);
    }
    static bool TryDecompress(Context& c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t& written)   // :172-186
    {
        const int st = snp_try_decompress(c.get(), in, n, out, cap, &written);
        if (st == SNP_ERR_OUTPUT_TOO_SMALL) { written = 0; return false; }
        ThrowFor(st, c);
        return true;
    }
    static std::vector<uint8_t> DecompressToArray(Context& c, const uint8_t* in, size_t n)         // :223-235,271-281
    {
        std::vector<uint8_t> buf(static_cast<size_t>(GetUncompressedLength(in, n)));
        size_t w = 0;
        if (!TryDecompress(c, in, n, buf.data(), buf.size(), w)) throw ArgumentException("Output buffer is too small.");
        buf.resize(w);
        return buf;
    }
};

}  // namespace Snappier
