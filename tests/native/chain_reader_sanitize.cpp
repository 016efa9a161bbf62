// Sanitizer harness for the chain reader (CPU only): compiled by tests/test_chain_reader.py with
// -fsanitize=address,undefined together with mcevidence_amd/csrc/chain_reader.cpp and run as a process.
// Exercises mmap edges (no trailing newline, empty file, file ending in a partial token), every thread
// count against ragged/garbage input, and repeated open/close.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mcechains.h"

static std::string write_file(const std::string& dir, const char* name, const std::string& body)
{
    const std::string p = dir + "/" + name;
    FILE* f = std::fopen(p.c_str(), "wb");
    if (!f) { std::perror("fopen"); std::exit(2); }
    std::fwrite(body.data(), 1, body.size(), f);
    std::fclose(f);
    return p;
}

static int check(const std::string& path, int nthreads, long want_rows, long want_cols, int want_rc)
{
    void* h = nullptr;
    int64_t nr = -1, nc = -1;
    int rc = mce_chain_open(path.c_str(), nthreads, &h, &nr, &nc);
    if (rc != MCC_OK) return rc == want_rc ? 0 : 1;
    std::vector<double> out((size_t)(nr * nc) + 1, -777.0);
    rc = mce_chain_read(h, out.data());
    mce_chain_close(h);
    if (rc != want_rc) { std::fprintf(stderr, "%s: rc %d, wanted %d (%s)\n", path.c_str(), rc, want_rc, mce_chain_last_error()); return 1; }
    if (rc == MCC_OK && (nr != want_rows || nc != want_cols)) { std::fprintf(stderr, "%s: shape %lld x %lld\n", path.c_str(), (long long)nr, (long long)nc); return 1; }
    if (out.back() != -777.0) { std::fprintf(stderr, "%s: wrote past the buffer\n", path.c_str()); return 1; }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    int bad = 0;
    std::string big;
    for (int i = 0; i < 30000; ++i) {
        char line[128];
        std::snprintf(line, sizeof(line), "%d %.8E\t%.17g  -%d.5e-%d\n", i % 7 + 1, i * 1.25e-3, 1.0 / (i + 1), i, i % 300);
        big += line;
        if (i % 1000 == 0) big += "# comment\n\n";
    }
    const std::string pbig = write_file(dir, "big.txt", big);
    const std::string pnonl = write_file(dir, "nonl.txt", "1 2 3\n4 5 6");
    const std::string pempty = write_file(dir, "empty.txt", "");
    const std::string pcomm = write_file(dir, "comm.txt", "#a\n#b");
    const std::string ptrunc = write_file(dir, "trunc.txt", "1.5 2.5\n3.5 4.5e");
    const std::string pragged = write_file(dir, "ragged.txt", big + "1 2\n");
    const std::string plong = write_file(dir, "long.txt", std::string(5000, '9') + " 1\n");
    const std::string pcr = write_file(dir, "cr.txt", "1 2\r3 4\r\n5 6\n");
    for (int nt : {0, 1, 2, 3, 7, 16, 64, 1000}) {
        bad += check(pbig, nt, 30000, 4, MCC_OK);
        bad += check(pnonl, nt, 2, 3, MCC_OK);
        bad += check(pempty, nt, 0, 0, MCC_OK);
        bad += check(pcomm, nt, 0, 0, MCC_OK);
        bad += check(ptrunc, nt, 0, 0, MCC_ERR_PARSE);
        bad += check(pragged, nt, 0, 0, MCC_ERR_RAGGED);
        bad += check(plong, nt, 0, 0, MCC_ERR_PARSE);          // 5000-digit token: refused, not overflowed
        bad += check(pcr, nt, 3, 2, MCC_OK);
    }
    bad += check(dir + "/does_not_exist.txt", 0, 0, 0, MCC_ERR_IO);
    bad += check(dir, 0, 0, 0, MCC_ERR_IO);
    double v = 0;
    bad += mce_chain_parse_token("1e", 2, &v) != MCC_ERR_PARSE;
    bad += mce_chain_parse_token("", 0, &v) != MCC_ERR_PARSE;
    bad += mce_chain_parse_token(nullptr, 0, &v) != MCC_ERR_INVALID;
    std::printf(bad ? "FAILED %d\n" : "OK\n", bad);
    return bad != 0;
}
