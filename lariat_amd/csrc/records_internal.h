// records_internal.h — one BAM record as records.cpp derives it from a result (bamwriter.go AppendBam), before it is rendered
// as a text line (lh_records_text) or encoded as a BAM record (bamfile.cpp).  Internal to the library.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>
#include "../../include/lariat_hip.h"

struct LhRecTag { char tag[2]; char type; std::string z; int32_t i; };   // type 'Z' or 'i'
struct LhRec {
    const char* name = nullptr; size_t name_len = 0;
    int flags = 0, mapq = 0;
    int32_t rid = -1, mrid = -1;        // -1: '*'
    int64_t pos = -1, mpos = -1, tlen = 0;
    std::vector<uint32_t> cig_len; std::vector<char> cig_op;   // op letters M I D S H
    std::string seq, qual;              // empty: '*'
    std::vector<LhRecTag> tags; size_t n_tags = 0;   // tags[0..n_tags) are in use (the vector only grows: strings keep their capacity)
    LhRecTag& tag(const char* t, char type) {
        if (n_tags == tags.size()) tags.emplace_back();
        LhRecTag& g = tags[n_tags++];
        g.tag[0] = t[0]; g.tag[1] = t[1]; g.type = type; g.z.clear(); g.i = 0;
        return g;
    }
};
// every record of the batch in the reference's order (DoDumpToBam), ranges of pairs handled by host threads: sink(thread, record)
// is called from thread `thread` (0 .. n_threads-1) in order; threads own consecutive ranges of pairs.
int lh_records_visit_(const lh_result* res, const lh_ingest_batch* in, int32_t n_contigs, const char* const* contig_names, int32_t flags /* LH_REC_* */, int* n_threads,
                      const std::function<void(int)>& begin_thread, const std::function<void(int, const LhRec&)>& sink);
