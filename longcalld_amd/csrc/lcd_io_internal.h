// lcd_io.cpp's helpers used by lcd_host.cpp (same library; not part of the C ABI)
#pragma once
#include <stdint.h>
#include <utility>
#include <vector>
// The BGZF blocks a region's records can lie in, through the .bai: `image` = those whole blocks back to back (a valid input of lcd_bgzf_inflate_dev), `ranges` = the
// merged index chunks as [begin, end) offsets of the image's INFLATED stream, in file order.  tid / tlen / n_ref from the BAM header.
struct LcdRegionImage { std::vector<uint8_t> image; std::vector<std::pair<uint64_t, uint64_t>> ranges; int tid = -1, n_ref = 0; int64_t tlen = 0; };
int lcd_io_region_image(const char *bam_path, const char *bai_path, const char *chrom, int64_t reg_beg, int64_t reg_end, LcdRegionImage &out);
