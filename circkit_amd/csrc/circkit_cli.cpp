// circkit_cli.cpp -- `circkit canonicalize` / `circkit uniq` on the MI355X path.
//
// Host-side mirror of the reference's two subcommand drivers (same flags, defaults, output format, exit
// behaviour):
//   src/commands.rs:112-149   Canonicalize{input,output,threads}, Uniq{input,output,canonicalize,table,threads}
//   src/canonicalize.rs:7-51  reader -> worker(normalize + canonicalize) -> writer(">head\nSEQ\n")
//   src/uniq.rs:15-88         same worker; xxh3 -> first-seen map; kept record or (id, duplicate_id) table row
//   src/utils.rs:9-84         input (file | stdin, compression sniffed by magic bytes), output (compression by
//                             extension), table writer (tab for .tsv, comma otherwise)
// The per-record work runs on the GPU through the C ABI (include/circkit.h); this file only streams text in,
// packs CSR batches (fasta_host.h) and writes text out.  There is no CPU compute path: without a usable GPU the
// program fails like any other I/O error.  Compressed streams are piped through the system's gzip / bzip2 / xz /
// zstd binaries.
#include <dlfcn.h>
#include <errno.h>
#include <math.h>
#include <sched.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <new>
#include <memory>
#include <mutex>
#include <climits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/circkit.h"
#include "fasta_host.h"

namespace {

struct Options {
    std::string cmd, input, output, table;
    bool has_input = false, has_output = false, has_table = false, canonicalize = false;
    int threads = 0, device = 0;
    bool has_bases = false, has_percent = false;        // rotate
    long long bases = 0;
    double percent = 0.0;
};

[[noreturn]] void die(const std::string& msg)
{
    fprintf(stderr, "Error: %s\n", msg.c_str());     // anyhow's `Error: ...` on stderr, exit code 1 (src/main.rs:13,39)
    fflush(stderr);
    _exit(1);      // not exit(): other pipeline threads may be blocked on objects that static destruction would tear down
}

void usage(FILE* f)
{
    fprintf(f,
            "circkit (MI355X build)\n\nUSAGE:\n"
            "    circkit canonicalize [INPUT] [-o <OUTPUT>] [-t <THREADS>]\n"
            "    circkit uniq [INPUT] [-o <OUTPUT>] [-c|--canonicalize] [--table <TABLE>] [-t <THREADS>]\n"
            "    circkit rotate [INPUT] [-o <OUTPUT>] (-b|--bases <N> | -p|--percent <FRACTION>)\n"
            "    circkit cat [INPUT] [-o <OUTPUT>]\n"
            "    circkit decat [INPUT] [-o <OUTPUT>]\n\n"
            "    INPUT   FASTA file, may be gzip, bzip, xz, or zstd compressed [default: stdin]\n"
            "    -o      output FASTA path; .gz/.bz2/.xz/.zst compress [default: stdout]\n"
            "    -c      uniq: output canonicalized sequences (aliases --norm --canon)\n"
            "    --table uniq: CSV (TSV for .tsv) of id,duplicate_id\n"
            "    -t      host parser threads [default: logical cores, at most 16]\n"
            "    --device <N>  GPU index [default: 0]\n"
            "    -b      rotate: bases to rotate by (positive: to the right, negative: to the left)\n"
            "    -p      rotate: fraction of the sequence length to rotate by, e.g. 0.5\n"
            "  canonicalize and uniq run on the GPU; rotate, cat and decat are byte copies done on the host.\n");
}

// clap's accepted forms for the arms this binary provides (src/commands.rs:6-14,93-180; clap 3 derive defaults): long options
// as `--name value` or `--name=value`; short options as `-o value`, `-ovalue`, `-o=value`; short flags and one trailing valued
// short combined in a cluster (`-ct4`); `--` ends the options; the global `-v/--verbose` and `-q/--quiet` of clap_verbosity_flag
// (repeatable, `global = true`: before or after the subcommand) -- they set the reference's log level and are no-ops here (this
// binary logs nothing); `-h/--help`, `-V/--version`.  Errors are clap's: a message on stderr, exit code 2.
[[noreturn]] void arg_error(const std::string& msg)
{
    fprintf(stderr, "error: %s\n\nFor more information try --help\n", msg.c_str());
    exit(2);
}
long long parse_int(const std::string& v, const char* what, long long lo, long long hi)
{
    char* end = nullptr;
    errno = 0;
    const long long x = strtoll(v.c_str(), &end, 10);
    if (v.empty() || *end || errno || x < lo || x > hi) arg_error("invalid value '" + v + "' for '" + what + "'");
    return x;
}
bool is_verbosity_arg(const std::string& a)
{
    if (a == "--verbose" || a == "--quiet") return true;
    if (a.size() < 2 || a[0] != '-' || a[1] == '-') return false;
    for (size_t k = 1; k < a.size(); ++k) if (a[k] != 'v' && a[k] != 'q') return false;
    return true;
}

Options parse_args(int argc, char** argv)
{
    Options o;
    int i = 1;
    while (i < argc && is_verbosity_arg(argv[i])) ++i;          // global flags in front of the subcommand
    if (i >= argc) { usage(stderr); exit(2); }
    o.cmd = argv[i++];
    if (o.cmd == "-h" || o.cmd == "--help" || o.cmd == "help") { usage(stdout); exit(0); }
    if (o.cmd == "-V" || o.cmd == "--version") { printf("circkit 0.1.0 (MI355X build)\n"); exit(0); }
    if (o.cmd != "canonicalize" && o.cmd != "uniq" && o.cmd != "rotate" && o.cmd != "cat" && o.cmd != "decat") {
        fprintf(stderr, "error: unrecognized subcommand '%s' (this build provides canonicalize, uniq, rotate, cat, decat)\n", o.cmd.c_str());
        exit(2);
    }
    const bool gpu_cmd = o.cmd == "canonicalize" || o.cmd == "uniq", uniq = o.cmd == "uniq", rotate = o.cmd == "rotate";
    // the valued options of this arm, by canonical long name
    auto set_value = [&](const std::string& name, const std::string& v) {
        if (name == "output") { o.output = v; o.has_output = true; }
        else if (name == "threads") o.threads = (int)parse_int(v, "--threads <THREADS>", 0, 2147483647ll);
        else if (name == "device") o.device = (int)parse_int(v, "--device <N>", 0, 1 << 20);
        else if (name == "table") { o.table = v; o.has_table = true; }
        else if (name == "bases") { o.bases = parse_int(v, "--bases <BASES>", LLONG_MIN, LLONG_MAX); o.has_bases = true; }      // negative values allowed (src/commands.rs:164)
        else {
            char* end = nullptr;
            o.percent = strtod(v.c_str(), &end);
            if (v.empty() || *end) arg_error("invalid value '" + v + "' for '--percent <PERCENT>'");
            o.has_percent = true;
        }
    };
    auto long_takes_value = [&](const std::string& n) {
        return n == "output" || (gpu_cmd && (n == "threads" || n == "device")) || (uniq && n == "table") || (rotate && (n == "bases" || n == "percent"));
    };
    auto short_name = [&](char ch) -> const char* {             // valued shorts
        if (ch == 'o') return "output";
        if (gpu_cmd && ch == 't') return "threads";
        if (rotate && ch == 'b') return "bases";
        if (rotate && ch == 'p') return "percent";
        return nullptr;
    };
    auto next_value = [&](const std::string& shown) -> std::string {
        if (i + 1 >= argc) arg_error("the argument '" + shown + "' requires a value but none was supplied");
        return argv[++i];
    };
    bool only_positional = false;
    for (; i < argc; ++i) {
        const std::string a = argv[i];
        if (only_positional || a.size() < 2 || a[0] != '-') {         // a positional ("-" included: a path like any other)
            if (o.has_input) arg_error("unexpected argument '" + a + "' found");
            o.input = a; o.has_input = true;
            continue;
        }
        if (a == "--") { only_positional = true; continue; }
        if (a[1] == '-') {
            const size_t eq = a.find('=');
            const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (long_takes_value(name)) set_value(name, eq == std::string::npos ? next_value(a) : a.substr(eq + 1));
            else if (eq != std::string::npos) arg_error("unexpected value '" + a.substr(eq + 1) + "' for '--" + name + "' found; no more were expected");
            else if (uniq && (name == "canonicalize" || name == "norm" || name == "canon")) o.canonicalize = true;
            else if (name == "verbose" || name == "quiet") {}
            else if (name == "help") { usage(stdout); exit(0); }
            else if (name == "version") { printf("circkit 0.1.0 (MI355X build)\n"); exit(0); }
            else arg_error("unexpected argument '" + a + "' found");
            continue;
        }
        for (size_t k = 1; k < a.size(); ++k) {                      // a cluster of shorts
            const char ch = a[k];
            if (const char* name = short_name(ch)) {
                std::string v;
                if (k + 1 < a.size()) { v = a.substr(k + 1); if (v[0] == '=') v = v.substr(1); }
                else v = next_value(std::string("-") + ch);
                set_value(name, v);
                break;
            }
            if (uniq && ch == 'c') o.canonicalize = true;
            else if (ch == 'v' || ch == 'q') {}
            else if (ch == 'h') { usage(stdout); exit(0); }
            else if (ch == 'V') { printf("circkit 0.1.0 (MI355X build)\n"); exit(0); }
            else arg_error(std::string("unexpected argument '-") + ch + "' found");
        }
    }
    if (o.has_bases && o.has_percent)           // src/commands.rs:164,170: the two flags exclude each other
        arg_error("the argument '--bases <BASES>' cannot be used with '--percent <PERCENT>'");
    return o;
}

std::string shell_quote(const std::string& s)
{
    std::string q = "'";
    for (char c : s) { if (c == '\'') q += "'\\''"; else q += c; }
    return q + "'";
}

// ---- zstd without the `zstd` binary: libzstd.so.1 (present wherever ROCm is) through dlopen, run as a filter in a
// forked child so that the pipeline keeps reading / writing a plain pipe.  Only the stable streaming ABI is used.
struct ZBufIn { const void* src; size_t size, pos; };
struct ZBufOut { void* dst; size_t size, pos; };
struct ZstdApi {
    void* (*createD)(); size_t (*freeD)(void*); size_t (*decompressStream)(void*, ZBufOut*, ZBufIn*);
    void* (*createC)(); size_t (*freeC)(void*); size_t (*initC)(void*, int);
    size_t (*compressStream)(void*, ZBufOut*, ZBufIn*); size_t (*endStream)(void*, ZBufOut*);
    unsigned (*isError)(size_t);
    bool load()
    {
        void* h = dlopen("libzstd.so.1", RTLD_NOW);
        if (!h) return false;
#define ZS(f, n) f = (decltype(f))dlsym(h, n); if (!f) return false
        ZS(createD, "ZSTD_createDStream"); ZS(freeD, "ZSTD_freeDStream"); ZS(decompressStream, "ZSTD_decompressStream");
        ZS(createC, "ZSTD_createCStream"); ZS(freeC, "ZSTD_freeCStream"); ZS(initC, "ZSTD_initCStream");
        ZS(compressStream, "ZSTD_compressStream"); ZS(endStream, "ZSTD_endStream"); ZS(isError, "ZSTD_isError");
#undef ZS
        return true;
    }
};
bool write_all(int fd, const void* p, size_t n)
{
    const char* c = (const char*)p;
    while (n) {
        const ssize_t w = write(fd, c, n);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        c += w; n -= (size_t)w;
    }
    return true;
}
// child body: in_fd -> (de)compress -> out_fd; exit code 0 on success
[[noreturn]] void zstd_filter(const ZstdApi& z, int in_fd, int out_fd, bool compress, int level)
{
    std::vector<char> ib(1 << 20), ob(1 << 20);
    void* st = compress ? z.createC() : z.createD();
    if (!st || (compress && z.isError(z.initC(st, level)))) _exit(3);
    bool mid_frame = false;
    for (;;) {
        const ssize_t got = read(in_fd, ib.data(), ib.size());
        if (got < 0) { if (errno == EINTR) continue; _exit(4); }
        if (got == 0) break;
        ZBufIn in{ ib.data(), (size_t)got, 0 };
        while (in.pos < in.size) {
            ZBufOut out{ ob.data(), ob.size(), 0 };
            const size_t rc = compress ? z.compressStream(st, &out, &in) : z.decompressStream(st, &out, &in);
            if (z.isError(rc)) _exit(5);
            mid_frame = !compress && rc != 0;
            if (out.pos && !write_all(out_fd, ob.data(), out.pos)) _exit(6);
        }
    }
    if (compress) {
        for (;;) {
            ZBufOut out{ ob.data(), ob.size(), 0 };
            const size_t left = z.endStream(st, &out);
            if (z.isError(left)) _exit(5);
            if (out.pos && !write_all(out_fd, ob.data(), out.pos)) _exit(6);
            if (left == 0) break;
        }
    } else if (mid_frame) _exit(7);                    // truncated input
    _exit(0);
}
// forks the filter between `fd` and a new pipe; returns the pipe end the parent uses (read end for decompression,
// write end for compression) or -1 when libzstd is not there
int spawn_zstd(int fd, bool compress, int level, pid_t* child)
{
    ZstdApi z;
    if (!z.load()) return -1;
    int pp[2];
    if (pipe(pp) != 0) return -1;
    fflush(stdout); fflush(stderr);
    const pid_t pid = fork();
    if (pid < 0) { close(pp[0]); close(pp[1]); return -1; }
    if (pid == 0) {
        if (compress) { close(pp[1]); zstd_filter(z, pp[0], fd, true, level); }
        close(pp[0]); zstd_filter(z, fd, pp[1], false, 0);
    }
    *child = pid;
    close(fd);
    if (compress) { close(pp[0]); return pp[1]; }
    close(pp[1]); return pp[0];
}
bool reap_zstd(pid_t child)
{
    int st = 0;
    while (waitpid(child, &st, 0) < 0 && errno == EINTR) {}
    return WIFEXITED(st) && WEXITSTATUS(st) == 0;
}

// src/utils.rs:9-27: open file or stdin, sniff the compression format by magic bytes (niffler)
struct Input {
    FILE* f = nullptr;
    bool piped = false, zstd_child = false;
    pid_t zpid = -1;
    pid_t filter = -1, feeder = -1;    // stdin behind a decompressor: the filter process and (unseekable stdin) the one that feeds it
    const uint8_t* map = nullptr;      // regular uncompressed file: mapped, parsed in place (no read copy)
    size_t map_len = 0;
    uint8_t prefix[8];                 // uncompressed stdin: the bytes the sniffer consumed, handed out first
    size_t prefix_len = 0, prefix_pos = 0;
};
// fread() that hands out the sniffed prefix first
size_t in_read(Input& in, uint8_t* dst, size_t n)
{
    size_t done = 0;
    while (in.prefix_pos < in.prefix_len && done < n) dst[done++] = in.prefix[in.prefix_pos++];
    if (done < n) done += fread(dst + done, 1, n - done, in.f);
    return done;
}
void close_input(Input& in)
{
    if (in.piped) { if (pclose(in.f) != 0) die("the input decompressor failed (is it installed?)"); }
    else if (in.zstd_child) { fclose(in.f); if (!reap_zstd(in.zpid)) die("zstd decompression of the input failed"); }
    else if (in.filter > 0) {
        fclose(in.f);
        const bool ok = reap_zstd(in.filter);          // (waitpid + exit status 0)
        if (in.feeder > 0) (void)reap_zstd(in.feeder);
        if (!ok) die("the input decompressor failed (is it installed?)");
    } else if (in.f != stdin) fclose(in.f);
}
// child process running `sh -c cmd` with stdin = in_fd; returns the read end of its stdout
int spawn_filter(const char* cmd, int in_fd, pid_t* child)
{
    int pp[2];
    if (pipe(pp) != 0) return -1;
    fflush(stdout); fflush(stderr);
    const pid_t pid = fork();
    if (pid < 0) { close(pp[0]); close(pp[1]); return -1; }
    if (pid == 0) {
        dup2(in_fd, 0); dup2(pp[1], 1);
        close(pp[0]); close(pp[1]); if (in_fd != 0) close(in_fd);
        execl("/bin/sh", "sh", "-c", cmd, (char*)nullptr);
        _exit(127);
    }
    *child = pid;
    close(pp[1]);
    return pp[0];
}
// child process that replays `head` and then forwards the rest of fd 0; returns the read end
int spawn_feeder(const uint8_t* head, size_t n, pid_t* child)
{
    int pp[2];
    if (pipe(pp) != 0) return -1;
    fflush(stdout); fflush(stderr);
    const pid_t pid = fork();
    if (pid < 0) { close(pp[0]); close(pp[1]); return -1; }
    if (pid == 0) {
        close(pp[0]);
        static uint8_t buf[1 << 20];
        size_t have = n;
        memcpy(buf, head, n);
        for (;;) {
            for (size_t off = 0; off < have;) {
                const ssize_t w = write(pp[1], buf + off, have - off);
                if (w < 0) { if (errno == EINTR) continue; _exit(1); }
                off += (size_t)w;
            }
            ssize_t r;
            while ((r = read(0, buf, sizeof buf)) < 0 && errno == EINTR) {}
            if (r <= 0) _exit(r < 0);
            have = (size_t)r;
        }
    }
    *child = pid;
    close(pp[1]);
    return pp[0];
}

const char* sniff(const uint8_t* m, size_t n)
{
    if (n >= 2 && m[0] == 0x1f && m[1] == 0x8b) return "gzip";
    if (n >= 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h') return "bzip2";
    if (n >= 6 && m[0] == 0xfd && m[1] == '7' && m[2] == 'z' && m[3] == 'X' && m[4] == 'Z' && m[5] == 0) return "xz";
    if (n >= 4 && m[0] == 0x28 && m[1] == 0xb5 && m[2] == 0x2f && m[3] == 0xfd) return "zstd";
    return nullptr;
}

Input open_input(const Options& o)
{
    Input in;
    if (!o.has_input) {
        if (isatty(0)) die("No stdin detected. Did you mean to include a file argument?");     // src/utils.rs:18-20
        // niffler::send::get_reader wraps stdin as well (src/utils.rs:21-24): `circkit canonicalize < in.fasta.gz` works
        // there.  Sniff the first bytes straight from the descriptor (nothing is buffered in `stdin` yet).
        uint8_t magic[6];
        size_t got = 0;
        while (got < sizeof magic) {
            const ssize_t r = read(0, magic + got, sizeof magic - got);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            got += (size_t)r;
        }
        const char* tool = sniff(magic, got);
        in.f = stdin;
        if (!tool) { memcpy(in.prefix, magic, got); in.prefix_len = got; return in; }
        int src = 0;                                 // a redirected file: rewind; a pipe: a feeder child replays the sniffed bytes
        if (lseek(0, 0, SEEK_SET) != 0) {
            src = spawn_feeder(magic, got, &in.feeder);
            if (src < 0) die("could not start the stdin feeder");
        }
        int pfd = -1;
        if (!strcmp(tool, "zstd")) {
            pfd = spawn_zstd(src == 0 ? dup(0) : src, false, 0, &in.zpid);
            if (pfd >= 0) { in.f = fdopen(pfd, "rb"); in.zstd_child = true; return in; }
            if (src != 0) die("zstd decompression of stdin is not available");
        }
        pfd = spawn_filter((std::string(tool) + " -dc").c_str(), src, &in.filter);
        if (pfd < 0) die(std::string("could not run ") + tool + " to decompress the input");
        if (src != 0) close(src);
        in.f = fdopen(pfd, "rb");
        return in;
    }
    FILE* f = fopen(o.input.c_str(), "rb");
    if (!f) die(std::string(strerror(errno)) + " (os error " + std::to_string(errno) + ")");  // tests/canon_uniq.rs:9-16
    uint8_t magic[6];
    const size_t got = fread(magic, 1, sizeof magic, f);
    const char* tool = sniff(magic, got);
    if (!tool) {
        rewind(f);
        in.f = f;
        struct stat st;
        if (fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                in.map = (const uint8_t*)m;
                in.map_len = (size_t)st.st_size;
            }
        }
        return in;
    }
    if (!strcmp(tool, "zstd")) {                    // libzstd in a forked filter; the `zstd` binary only as a fallback
        const int rfd = dup(fileno(f));
        if (rfd >= 0 && lseek(rfd, 0, SEEK_SET) == 0) {
            const int pfd = spawn_zstd(rfd, false, 0, &in.zpid);
            if (pfd >= 0) { fclose(f); in.f = fdopen(pfd, "rb"); in.zstd_child = true; return in; }
            close(rfd);
        }
    }
    fclose(f);
    const std::string cmd = std::string(tool) + " -dc " + shell_quote(o.input);
    in.f = popen(cmd.c_str(), "r");
    if (!in.f) die(std::string("could not run ") + tool + " to decompress the input");
    in.piped = true;
    return in;
}

// src/utils.rs:29-72: compression by extension with the reference's levels (gz 6, bz2 9, xz 6, zst 1)
struct Output {
    FILE* f = nullptr;
    bool regular = false;              // an uncompressed regular file we created: chunks are assembled in parallel and written with pwrite
    bool piped = false, zstd_child = false;
    pid_t zpid = -1;
    std::vector<char> buf;
};

Output open_output(const Options& o)
{
    Output out;
    out.buf.resize(8 << 20);
    if (!o.has_output) { out.f = stdout; setvbuf(stdout, out.buf.data(), _IOFBF, out.buf.size()); return out; }
    const std::string& p = o.output;
    auto ends = [&](const char* e) { const size_t n = strlen(e); return p.size() >= n && p.compare(p.size() - n, n, e) == 0; };
    const char* tool = ends(".gz") ? "gzip -6" : ends(".bz2") ? "bzip2 -9" : ends(".xz") ? "xz -6" : ends(".zst") ? "zstd -1 -q" : nullptr;
    FILE* probe = fopen(p.c_str(), "wb");
    if (!probe) die("Could not create output file " + p + ". Are you sure it's not actually a directory?");   // src/utils.rs:46-49
    if (!tool) {
        out.f = probe; setvbuf(out.f, out.buf.data(), _IOFBF, out.buf.size());
        struct stat st;
        out.regular = fstat(fileno(probe), &st) == 0 && S_ISREG(st.st_mode) && !getenv("CIRCKIT_CLI_WRITEV_OUTPUT");
        return out;
    }
    if (ends(".zst")) {                              // level 1 (src/utils.rs:58-60) through libzstd, no `zstd` binary needed
        const int wfd = dup(fileno(probe));
        const int pfd = wfd >= 0 ? spawn_zstd(wfd, true, 1, &out.zpid) : -1;
        if (pfd >= 0) {
            fclose(probe);
            out.f = fdopen(pfd, "wb");
            setvbuf(out.f, out.buf.data(), _IOFBF, out.buf.size());
            out.zstd_child = true;
            return out;
        }
        if (wfd >= 0) close(wfd);
    }
    fclose(probe);
    const std::string cmd = std::string(tool) + " -c > " + shell_quote(p);
    out.f = popen(cmd.c_str(), "w");
    if (!out.f) die(std::string("could not run `") + tool + "` to compress the output");
    out.piped = true;
    return out;
}

void close_output(Output& out)
{
    if (!out.f) return;
    if (fflush(out.f) != 0) die("failed to write output");
    if (out.piped) { if (pclose(out.f) != 0) die("the output compressor failed (is it installed?)"); }
    else if (out.zstd_child) { fclose(out.f); if (!reap_zstd(out.zpid)) die("zstd compression of the output failed"); }
    else if (out.f != stdout) fclose(out.f);
    out.f = nullptr;
}

void check(circkit_ctx* ctx, int rc)
{
    if (rc != CIRCKIT_OK) die(std::string("GPU path failed: ") + circkit_last_error(ctx));
}

// CIRCKIT_CLI_TIMING=1: busy seconds per pipeline stage on stderr (stages overlap; wall is what counts)
struct Busy {
    std::atomic<long> ns{ 0 };
    struct Scope {
        Busy& b; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        explicit Scope(Busy& bb) : b(bb) {}
        ~Scope() { b.ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
    };
    double s() const { return ns.load() * 1e-9; }
};
Busy g_read, g_parse, g_gpu, g_write, g_emit, g_pwrite;

// ---- the pipeline ------------------------------------------------------------------------------------------
// reader -> parser pool -> GPU -> writer over a ring of chunk slots.  Mirrors seq_io's parallel_fasta shape
// (one reader, `--threads` workers, results consumed in input order: src/canonicalize.rs:17-45) with the worker
// closure replaced by "parse + normalize + pack" on the host and ONE GPU batch call per chunk.
enum State { FREE, READ, PARSED, COMPUTED, ASSEMBLED };

struct Slot {
    State state = FREE;
    long seq = -1;                              // sequence number of the chunk this slot currently holds
    std::vector<uint8_t> own;                   // chunk text when the input is a stream
    const uint8_t* text = nullptr;              // -> own, or into the mapped input file
    size_t len = 0;
    bool first = false;
    ckhost::Batch batch;                        // batch.bytes and canon live in pinned memory: DMA both ways
    ckhost::ByteBuf canon;
    // a chunk is parsed in sub-ranges by several threads (SubBatch: private buffers), then placed into `batch` as ONE CSR
    struct SubBatch { ckhost::Batch b; size_t start = 0; uint64_t rec0 = 0, byte0 = 0; };
    std::vector<std::unique_ptr<SubBatch>> sub;
    int n_sub = 0;
    std::atomic<int> parse_left{ 0 }, place_left{ 0 };
    std::vector<uint64_t> hash, first_seen;
    uint64_t base = 0;
    // positioned output (regular output file): byte offset of every record inside the chunk's window of the file
    std::vector<uint64_t> ooff;                 // n + 1; a dropped record (uniq) has size 0
    std::vector<uint8_t> obuf;                  // the chunk's output text, assembled by the emit workers
    std::atomic<int> emit_left{ 0 };            // emit jobs of this chunk still running
};

struct Pipeline {
    static constexpr int MAX_K = 32;
    static int K;                               // chunk slots in flight
    static size_t CHUNK;                        // text bytes per chunk (one GPU batch)
    Slot slot[MAX_K];
    std::mutex m;
    std::condition_variable cv;
    long n_chunks = -1;                         // set by the reader at EOF
    long next_parse = 0;                        // next chunk sequence number a parser may take

    void set(long seq, State st)
    {
        { std::lock_guard<std::mutex> g(m); slot[seq % K].state = st; if (st == READ) slot[seq % K].seq = seq; }
        cv.notify_all();
    }
    // waits until chunk `seq` (not an older tenant of the same slot) reaches `st`; false when the stream ended
    // before that chunk.  FREE is waited for by the reader only: any FREE slot will do.
    bool wait(long seq, State st)
    {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] {
            const Slot& s = slot[seq % K];
            return (s.state == st && (st == FREE || s.seq == seq)) || (n_chunks >= 0 && seq >= n_chunks);
        });
        return !(n_chunks >= 0 && seq >= n_chunks);
    }
};


int Pipeline::K = 6;
size_t Pipeline::CHUNK = 64u << 20;

// rotate / cat / decat (src/rotate.rs:9-50, src/concatenate.rs:10-54): per-record byte copies of full_seq() -- the
// sequence lines joined, nothing normalized -- done on the host; there is nothing for a GPU to win on a memcpy that
// starts and ends in host memory.
int run_host_edit(const Options& opt, Input& in, Output& out)
{
    if (opt.cmd == "rotate" && ((opt.has_bases && opt.bases == 0) || (opt.has_percent && opt.percent == 0.0)))
        die("Rotation by 0 is not allowed");                                                // src/rotate.rs:20-22
    std::vector<uint8_t> owned;
    const uint8_t* text = in.map;
    size_t len = in.map_len;
    if (!text) {
        uint8_t buf[1 << 16];
        size_t got;
        while ((got = in_read(in, buf, sizeof buf)) > 0) owned.insert(owned.end(), buf, buf + got);
        text = owned.data();
        len = owned.size();
    }
    ckhost::Batch b;
    std::string err;
    size_t used = 0;
    // a reader error ends the reference's `while let Some(Ok(record))` loop silently; so does it here
    if (len && ckhost::parse_chunk(text, len, true, true, b, &used, err)) {
        std::string seq, line;
        for (size_t i = 0; i < b.n(); ++i) {
            // full_seq(): the record's lines without their terminators (\n, and a \r before it)
            seq.clear();
            const uint8_t* p = text + b.raw[i].off;
            const size_t n = b.raw[i].len;
            for (size_t k = 0; k < n;) {
                const uint8_t* nl = (const uint8_t*)memchr(p + k, '\n', n - k);
                size_t e = nl ? (size_t)(nl - p) : n;
                const size_t next = nl ? e + 1 : n;
                if (e > k && p[e - 1] == '\r') --e;
                seq.append((const char*)p + k, e - k);
                k = next;
            }
            line.assign(">");
            line.append((const char*)text + b.head[i].off, b.head[i].len);
            line += '\n';
            if (opt.cmd == "cat") { line += seq; line += seq; }
            else if (opt.cmd == "decat") line.append(seq, 0, seq.size() / 2);
            else {
                if (!opt.has_bases && !opt.has_percent) {                                  // src/rotate.rs:29 `expect`
                    fprintf(stderr, "Must provide either --bases or --percent\n");
                    _exit(101);
                }
                if (seq.empty()) {                                                         // `% 0` panics in the reference
                    fprintf(stderr, "attempt to calculate the remainder with a divisor of zero\n");
                    _exit(101);
                }
                const long long start = opt.has_percent ? (long long)floor((double)seq.size() * opt.percent) : opt.bases;
                const size_t L = seq.size();
                const size_t at = start >= 0 ? L - (size_t)((unsigned long long)start % L)        // src/rotate.rs:37-40
                                             : (size_t)((0ULL - (unsigned long long)start) % L);
                line.append(seq, at, std::string::npos);
                line.append(seq, 0, at);
            }
            line += '\n';
            if (fwrite(line.data(), 1, line.size(), out.f) != line.size()) die("failed to write output");
        }
    }
    close_output(out);
    close_input(in);
    return 0;
}

// The reference's default is num_cpus::get() (src/commands.rs:120-123): the CPUs this process may run on, and under a
// cgroup CPU quota no more than the quota allows -- a container that shows 256 CPUs but grants 16 runs 64 threads slower
// than 16 (measured on the GPU boxes: 2.5-3.2 against 3.8 M records/s).
int default_threads()
{
    long n = (long)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = (atol(q) + period - 1) / period;
            if (quota >= 1 && quota < n) n = quota;
        }
        fclose(f);
    }
    return n < 1 ? 1 : (int)n;
}

}  // namespace

// ---- the supervisor ---------------------------------------------------------------------------------------------------
// When the last byte is written and every file is closed, the process still owns 768 MB of pinned buffers, the mapping of
// the input and a GPU context, and the kernel takes them apart before anyone waiting for the process is told that it has ended:
// 0.3 s for a 5 GB input, whatever the exit path (measured: `_exit` right behind the last close; tools/cli_start_probe.sh --
// an empty input leaves main() after 0.10 s and the shell sees 0.13-0.24).  So the GPU arms run in a WORKER process, and the
// process the caller started is a supervisor that owns nothing: it exits with the worker's status as soon as the worker
// reports, over a pipe, that all output is written and closed -- or, if the worker ends without that report (an error exit, a
// signal), with the worker's own wait status.  Signals sent to the supervisor are passed on.  CIRCKIT_CLI_NO_SUPERVISOR=1: one
// process (debuggers, sanitizers, leak checkers).
int g_done_fd = -1;                  // worker: where to report (-1: no supervisor)
pid_t g_worker = -1;                 // supervisor: whom to pass signals on to
void pass_signal(int sig) { if (g_worker > 0) kill(g_worker, sig); }
void report_done()                   // worker, after the last close
{
    if (g_done_fd < 0) return;
    (void)close(STDOUT_FILENO);      // (a reader of our stdout must not wait for the teardown either,
    (void)close(STDERR_FILENO);      //  nor one that collects our messages: nothing is said after this point)
    const unsigned char ok = 0;
    ssize_t w;
    do w = write(g_done_fd, &ok, 1); while (w < 0 && errno == EINTR);
}
void supervise()                     // returns in the worker; the supervisor never returns
{
    if (getenv("CIRCKIT_CLI_NO_SUPERVISOR")) return;
    int fds[2];
    if (pipe2(fds, O_CLOEXEC) != 0) return;
    fflush(nullptr);
    const pid_t pid = fork();
    if (pid < 0) { close(fds[0]); close(fds[1]); return; }             // no second process: carry on alone
    if (pid == 0) { close(fds[0]); g_done_fd = fds[1]; return; }
    close(fds[1]);
    g_worker = pid;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = pass_signal;
    for (int sig : { SIGINT, SIGTERM, SIGHUP, SIGQUIT }) sigaction(sig, &sa, nullptr);
    (void)close(STDIN_FILENO);                                          // the worker's streams are the worker's
    (void)close(STDOUT_FILENO);
    unsigned char st = 0;
    ssize_t r;
    do r = read(fds[0], &st, 1); while (r < 0 && errno == EINTR);
    if (r == 1) _exit(st);                                              // everything is written and closed
    int ws = 0;
    pid_t w;
    do w = waitpid(pid, &ws, 0); while (w < 0 && errno == EINTR);
    if (w == pid && WIFEXITED(ws)) _exit(WEXITSTATUS(ws));
    if (w == pid && WIFSIGNALED(ws)) { signal(WTERMSIG(ws), SIG_DFL); kill(getpid(), WTERMSIG(ws)); _exit(128 + WTERMSIG(ws)); }
    _exit(1);
}

int main(int argc, char** argv)
{
    const auto t_main = std::chrono::steady_clock::now();
    const Options opt = parse_args(argc, argv);          // (usage errors end here, in the only process there is)
    if (opt.cmd == "canonicalize" || opt.cmd == "uniq") supervise();
    Input in = open_input(opt);
    Output out = open_output(opt);
    if (opt.cmd == "rotate" || opt.cmd == "cat" || opt.cmd == "decat") return run_host_edit(opt, in, out);
    FILE* table = nullptr;
    char delim = ',';
    if (opt.has_table) {
        table = fopen(opt.table.c_str(), "wb");
        if (!table) die("Could not create output table.");                                 // src/utils.rs:83
        if (opt.table.size() >= 4 && opt.table.compare(opt.table.size() - 4, 4, ".tsv") == 0) delim = '\t';   // src/utils.rs:77-80
    }
    circkit_ctx* ctx = nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    double init_s = 0;
    // HIP start-up (0.25-0.35 s) belongs to the GPU thread: reading and the first parses need no device and run next to it
    std::mutex ctx_m;
    std::condition_variable ctx_cv;
    bool ctx_ready = false;
    const bool uniq = opt.cmd == "uniq";
    const bool want_bytes = !uniq || opt.canonicalize;
    int n_parsers = opt.threads > 0 ? opt.threads : default_threads();   // src/commands.rs:120-123
    if (n_parsers < 1) n_parsers = 1;
    if (n_parsers > 64) n_parsers = 64;          // (beyond that the six chunk slots in flight are the limit, not the threads)
    if (getenv("CIRCKIT_CLI_SLOTS")) { const int k = atoi(getenv("CIRCKIT_CLI_SLOTS")); if (k >= 2 && k <= Pipeline::MAX_K) Pipeline::K = k; }
    if (getenv("CIRCKIT_CLI_CHUNK_MB")) { const int m = atoi(getenv("CIRCKIT_CLI_CHUNK_MB")); if (m >= 1 && m <= 1024) Pipeline::CHUNK = (size_t)m << 20; }
    static Pipeline P;

    for (Slot& sl : P.slot) {
        sl.batch.bytes.alloc = circkit_host_alloc; sl.batch.bytes.release = circkit_host_free;
        sl.canon.alloc = circkit_host_alloc; sl.canon.release = circkit_host_free;
    }
    // last record start ("\n>") in buf[1, len): everything before it is complete.  0 = none.
    auto last_record_start = [](const uint8_t* buf, size_t len) -> size_t {
        size_t k = len;
        while (k > 1) {
            const void* g = memrchr(buf + 1, '>', k - 1);
            if (!g) return 0;
            const size_t gi = (const uint8_t*)g - buf;
            if (buf[gi - 1] == '\n') return gi;
            k = gi;
        }
        return 0;
    };
    // the parsers' job queue (stage 2 below); the reader fills it
    struct ParseJob { long seq; int k; int phase; };
    std::deque<ParseJob> pjobs;
    std::mutex pm;
    std::condition_variable pcv;
    long chunks_parsed = 0, chunks_total = -1;        // (under pm; chunks_total: set by the reader at EOF)
    auto dispatch_chunk = [&](long seq) {             // (reader thread, after the slot is READ)
        Slot& s = P.slot[seq % Pipeline::K];
        static const size_t sub_bytes = getenv("CIRCKIT_CLI_SUB_KB") && atoi(getenv("CIRCKIT_CLI_SUB_KB")) > 0 ? (size_t)atoi(getenv("CIRCKIT_CLI_SUB_KB")) << 10 : (size_t)2 << 20;
        int n = (int)(s.len / sub_bytes);
        n = n < 1 ? 1 : (n > n_parsers ? n_parsers : n);
        while ((int)s.sub.size() < n) s.sub.emplace_back(new Slot::SubBatch());
        s.n_sub = n; s.parse_left = n; s.place_left = n;
        { std::lock_guard<std::mutex> g(pm); for (int k = 0; k < n; ++k) pjobs.push_back(ParseJob{ seq, k, 0 }); }
        pcv.notify_all();
    };
    // ---- stage 1: reader.  Cuts every chunk at a record start so the parsers never see a partial record.
    std::thread reader([&] {
        long seq = 0;
        if (in.map) {                               // mapped file: chunks are slices of the mapping
            size_t pos = 0;
            do {
                P.wait(seq, FREE);
                Busy::Scope tb(g_read);
                Slot& s = P.slot[seq % Pipeline::K];
                size_t want = Pipeline::CHUNK, cut = 0;
                for (;;) {
                    if (pos + want >= in.map_len) { cut = in.map_len - pos; break; }
                    cut = last_record_start(in.map + pos, want);
                    if (cut) break;
                    want *= 2;                      // one record longer than the chunk
                }
                s.text = in.map + pos; s.len = cut; s.first = seq == 0;
                pos += cut;
                P.set(seq, READ);
                dispatch_chunk(seq);
                ++seq;
            } while (pos < in.map_len);
        } else {
            std::vector<uint8_t> carry;
            bool eof = false;
            while (!eof || !carry.empty()) {
                P.wait(seq, FREE);
                Busy::Scope tb(g_read);
                Slot& s = P.slot[seq % Pipeline::K];
                size_t have = carry.size();
                if (s.own.size() < have + Pipeline::CHUNK) s.own.resize(have + Pipeline::CHUNK);
                memcpy(s.own.data(), carry.data(), have);
                carry.clear();
                size_t cut = 0;
                for (;;) {
                    while (!eof && have < s.own.size()) {
                        const size_t got = in_read(in, s.own.data() + have, s.own.size() - have);
                        if (got == 0) { if (ferror(in.f)) die("failed to read the input"); eof = true; }
                        have += got;
                    }
                    if (eof) { cut = have; break; }
                    cut = last_record_start(s.own.data(), have);
                    if (cut) break;
                    s.own.resize(s.own.size() * 2);     // one record longer than the chunk: keep reading
                }
                carry.assign(s.own.begin() + cut, s.own.begin() + have);
                s.text = s.own.data(); s.len = cut; s.first = seq == 0;
                if (cut == 0 && eof && carry.empty() && seq > 0) break;
                P.set(seq, READ);
                dispatch_chunk(seq);
                ++seq;
            }
        }
        { std::lock_guard<std::mutex> g(P.m); P.n_chunks = seq; }
        P.cv.notify_all();
        { std::lock_guard<std::mutex> g(pm); chunks_total = seq; }
        pcv.notify_all();
        // (Handing the slots' page-locked buffers back from here while the last chunks drain -- unpinning the ring's 768 MB is
        // 0.12 s of the process's exit -- was measured: hipHostFree waits for the device and holds up the copies still queued;
        // the pipeline's tail grew by more than the exit shrank, 0.97-1.03 -> 1.16-1.23 s wall into /dev/null.)
    });

    // ---- stage 2: parsers (FASTA record boundaries + needletail normalize + CSR pack).  A chunk is cut into sub-ranges at
    // record starts and every sub-range is a job of its own (round 3: one thread per 64 MB chunk, 27 ms -- with six chunks in
    // flight that latency bounded the whole pipeline): phase A parses a sub-range into private buffers, the last A job of a
    // chunk lays the chunk's CSR out (prefix sums over the sub-ranges' record and byte counts), phase B copies every
    // sub-range's payload, offsets and spans to their place in the chunk's batch -- one contiguous CSR, as the GPU call wants.
    std::vector<std::thread> parsers;
    for (int t = 0; t < n_parsers; ++t)
        parsers.emplace_back([&] {
            for (;;) {
                ParseJob j;
                {
                    std::unique_lock<std::mutex> g(pm);
                    pcv.wait(g, [&] { return !pjobs.empty() || (chunks_total >= 0 && chunks_parsed >= chunks_total); });
                    if (pjobs.empty()) return;
                    j = pjobs.front(); pjobs.pop_front();
                }
                Busy::Scope tb(g_parse);
                Slot& s = P.slot[j.seq % Pipeline::K];
                Slot::SubBatch& sb = *s.sub[j.k];
                if (j.phase == 0) {
                    std::string err;
                    if (!ckhost::parse_sub_range(s.text, s.len, s.first, s.n_sub, j.k, sb.b, &sb.start, err)) die(err);
                    if (--s.parse_left == 0) {
                        // layout of the chunk's CSR; the payload buffer is page-locked: the device must be there first
                        uint64_t rec = 0, bytes = 0;
                        for (int k = 0; k < s.n_sub; ++k) { Slot::SubBatch& q = *s.sub[k]; q.rec0 = rec; q.byte0 = bytes; rec += q.b.n(); bytes += q.b.offsets[q.b.n()]; }
                        { std::unique_lock<std::mutex> g(ctx_m); ctx_cv.wait(g, [&] { return ctx_ready; }); }
                        s.batch.text = s.text;
                        s.batch.head.resize(rec); s.batch.raw.resize(rec); s.batch.offsets.resize(rec + 1);
                        s.batch.bytes.len = 0;
                        s.batch.bytes.reserve(bytes + 64);
                        s.batch.bytes.len = bytes + 64;
                        s.batch.offsets[rec] = bytes;
                        memset(s.batch.bytes.data() + bytes, 0, 64);
                        { std::lock_guard<std::mutex> g(pm); for (int k = 0; k < s.n_sub; ++k) pjobs.push_front(ParseJob{ j.seq, k, 1 }); }
                        pcv.notify_all();
                    }
                } else {
                    ckhost::place_sub_batch(sb.b, sb.start, sb.rec0, sb.byte0, s.batch);
                    if (--s.place_left == 0) {
                        { std::lock_guard<std::mutex> g(pm); ++chunks_parsed; }
                        pcv.notify_all();
                        P.set(j.seq, PARSED);
                    }
                }
            }
        });

    // ---- stage 3: GPU, in input order (one ctx, one thread): canonical bytes / hashes / first-seen
    // One context.  (CIRCKIT_CLI_CTXS=2, `canonicalize` only: two contexts on two threads taking alternate chunks, so that the last
    // copy-out of chunk i overlaps the first copy-in of chunk i + 1 -- measured, tools/cli_ctx_probe.sh: every call then takes
    // twice as long, 5 GB into /dev/null 0.78-0.82 -> 0.96-1.01 s; the link and host DRAM are shared, not idle.  `uniq` always
    // keeps one: its first-seen table is fed in input order.)
    int n_ctx = 1;
    if (getenv("CIRCKIT_CLI_CTXS")) { const int k = atoi(getenv("CIRCKIT_CLI_CTXS")); if (k >= 1 && k <= 2 && !uniq) n_ctx = k; }
    circkit_ctx* ctx2 = nullptr;
    auto gpu_stage = [&](int which) {
        circkit_ctx*& my = which == 0 ? ctx : ctx2;
        const int rc = circkit_ctx_create(opt.device, &my);
        if (which == 0) init_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        if (rc != CIRCKIT_OK) die(std::string("no usable MI355X GPU (device ") + std::to_string(opt.device) + "): " +
                                   (my ? circkit_last_error(my) : "hipGetDeviceCount failed") + "; there is no CPU fallback");
        { std::lock_guard<std::mutex> g(ctx_m); ctx_ready = true; }
        ctx_cv.notify_all();
        uint64_t base = 0;
        for (long seq = which; P.wait(seq, PARSED); seq += n_ctx) {
            Busy::Scope tb(g_gpu);
            Slot& s = P.slot[seq % Pipeline::K];
            const uint64_t n = s.batch.n();
            s.base = base;                          // (global record index: uniq only, one context)
            if (n) {
                const uint64_t total = s.batch.offsets[n];
                if (want_bytes) { s.canon.len = 0; s.canon.reserve(total + 64); }
                if (uniq) { s.hash.resize(n); s.first_seen.resize(n); }
                check(my, circkit_canonicalize_batch(my, s.batch.bytes.data(), s.batch.offsets.data(), n,
                                                     want_bytes ? s.canon.data() : nullptr, nullptr, nullptr,
                                                     uniq ? s.hash.data() : nullptr));
                if (uniq) check(my, circkit_uniq_first_seen(my, s.hash.data(), n, base, s.first_seen.data()));
                base += n;
            }
            P.set(seq, COMPUTED);
        }
        // (Giving the ring's page-locked memory back from here -- the device has nothing queued any more -- instead of leaving it
        // to the process's exit moves 0.12 s of unpinning from behind the last byte written to in front of the last join: no gain.)
    };
    std::thread gpu([&] { gpu_stage(0); });
    std::thread gpu2;
    if (n_ctx == 2) gpu2 = std::thread([&] { gpu_stage(1); });

    // ---- stage 4: writer, in input order (this thread) + emit workers.
    // Regular output file: the writer only lays the chunk out (a prefix sum of the record sizes), grows the file and maps
    // the chunk's window; the emit workers copy headers and sequences into the mapping in parallel -- page-cache pages
    // are allocated by whoever touches them, so this scales where write() calls from several threads serialise on the
    // inode lock (tried: 1.86 s instead of 1.35 s for 5 GB on tmpfs).  Pipes, stdout and compressed output keep the
    // single-threaded writev straight from the chunk text and the pinned canonical buffer.
    std::vector<std::string> ids;                     // uniq --table: id of every kept record ...
    std::vector<uint64_t> kept_slot;                  // ... found through global index -> slot in ids
    bool table_header = false;
    std::string line;
    std::vector<struct iovec> iov;
    static const char GT = '>', NL = '\n';
    if (fflush(out.f) != 0) die("failed to write output");
    const int ofd = fileno(out.f);
    auto flush_iov = [&] {
        size_t i = 0;
        while (i < iov.size()) {
            const int cnt = (int)(iov.size() - i < 1024 ? iov.size() - i : 1024);
            ssize_t w = writev(ofd, iov.data() + i, cnt);
            if (w < 0) { if (errno == EINTR) continue; die("failed to write output"); }
            while (w > 0 && i < iov.size()) {          // partial write: advance inside the vector
                if ((size_t)w >= iov[i].iov_len) { w -= (ssize_t)iov[i].iov_len; ++i; }
                else { iov[i].iov_base = (char*)iov[i].iov_base + w; iov[i].iov_len -= (size_t)w; w = 0; }
            }
        }
        iov.clear();
    };
    // Every sink gets its chunks as ONE assembled block per chunk (emit workers) and one stream of large write() calls:
    // round 2 kept that for regular files and fed pipes / stdout / compressors by writev straight from the chunk text --
    // 25M tiny iovecs for 5M records, each copied into the pipe on its own: 2.4 s for 5 GB into `| cat > /dev/null` against
    // 1.0 s into a tmpfs file (tools/cli_e2e_sinks.sh).  CIRCKIT_CLI_WRITEV_OUTPUT=1 keeps the old path for comparison.
    const bool mapped_out = !getenv("CIRCKIT_CLI_WRITEV_OUTPUT");
    {
        struct stat st;
        if (mapped_out && fstat(ofd, &st) == 0 && S_ISFIFO(st.st_mode)) {
#ifdef F_SETPIPE_SZ
            (void)fcntl(ofd, F_SETPIPE_SZ, 1 << 20);       // the most an unprivileged process may ask for by default: fewer, larger hand-overs
#endif
        }
    }
    const int n_emit = mapped_out ? (n_parsers > 1 ? n_parsers : 2) : 0;
    struct EmitJob { long seq; uint64_t r0, r1; };
    std::deque<EmitJob> jobs;
    std::mutex jm;
    std::condition_variable jcv;
    bool jobs_done = false;
    std::vector<std::thread> emitters;
    for (int t = 0; t < n_emit; ++t)
        emitters.emplace_back([&] {
            for (;;) {
                EmitJob j;
                {
                    std::unique_lock<std::mutex> g(jm);
                    jcv.wait(g, [&] { return !jobs.empty() || jobs_done; });
                    if (jobs.empty()) return;
                    j = jobs.front(); jobs.pop_front();
                }
                Busy::Scope tb(g_emit);
                Slot& s = P.slot[j.seq % Pipeline::K];
                const ckhost::Batch& b = s.batch;
                uint8_t* base = s.obuf.data();
                for (uint64_t i = j.r0; i < j.r1; ++i) {
                    if (s.ooff[i + 1] == s.ooff[i]) continue;                  // dropped by uniq
                    uint8_t* d = base + s.ooff[i];
                    const ckhost::Span h = b.head[i];
                    *d++ = '>';
                    memcpy(d, s.text + h.off, h.len); d += h.len;
                    *d++ = '\n';
                    if (want_bytes) { const size_t L = (size_t)(b.offsets[i + 1] - b.offsets[i]); memcpy(d, s.canon.data() + b.offsets[i], L); d += L; }
                    else { memcpy(d, s.text + b.raw[i].off, b.raw[i].len); d += b.raw[i].len; }
                    *d = '\n';
                }
                if (--s.emit_left == 0) {
                    // The chunk's text is not needed again: drop its pages from the input mapping here, chunk by chunk and
                    // next to the other workers, instead of all 1.2M of a 5 GB file at exit (0.4 s of a 1.6 s run went
                    // into that teardown).  The page cache keeps the data; only this process's page tables shrink.
                    if (in.map && s.text >= in.map && s.text < in.map + in.map_len) {
                        const uintptr_t p0 = ((uintptr_t)s.text + 4095) & ~(uintptr_t)4095, p1 = (uintptr_t)(s.text + s.len) & ~(uintptr_t)4095;
                        if (p1 > p0) (void)madvise((void*)p0, p1 - p0, MADV_DONTNEED);
                    }
                    P.set(j.seq, ASSEMBLED);          // the writer thread puts it into the file
                }
            }
        });
    // chunks go into the file in order, by ONE thread: one stream of large write() calls is what the kernel's
    // per-inode lock allows anyway (positioned writes from the workers themselves: 3.1 s of pwrite for 1.6 s of wall;
    // workers filling shared mappings of the file, pages allocated by faults or by madvise(MADV_POPULATE_WRITE): the
    // allocations contend inside the kernel, 14-260 s of summed worker time for the same 1.0 s of pipeline at best)
    // ...by a thread that does nothing else: the write() stream is what a file sink waits for (5 GB: 0.8 s of it), and the
    // layout of the next chunk (a prefix sum over its records, 0.1 s in all) used to sit in the same thread between two writes.
    std::thread file_writer;
    if (mapped_out)
        file_writer = std::thread([&] {
            for (long wseq = 0; P.wait(wseq, ASSEMBLED); ++wseq) {
                Busy::Scope tw(g_pwrite);
                Slot& s = P.slot[wseq % Pipeline::K];
                const uint64_t total = s.ooff[s.batch.n()];
                for (uint64_t done = 0; done < total;) {
                    const ssize_t w = write(ofd, s.obuf.data() + done, (size_t)(total - done));
                    if (w < 0) { if (errno == EINTR) continue; die("failed to write output"); }
                    done += (uint64_t)w;
                }
                P.set(wseq, FREE);
            }
        });
    uint64_t file_size = 0;
    for (long seq = 0; P.wait(seq, COMPUTED); ++seq) {
        Busy::Scope tb(g_write);
        Slot& s = P.slot[seq % Pipeline::K];
        const ckhost::Batch& b = s.batch;
        const uint8_t* text = s.text;
        const uint64_t n = b.n();
        if (uniq && table) kept_slot.resize(s.base + n, ~0ull);
        if (mapped_out) s.ooff.resize(n + 1);
        uint64_t at = 0;
        for (uint64_t i = 0; i < n; ++i) {
            const ckhost::Span h = b.head[i];
            if (mapped_out) s.ooff[i] = at;
            if (uniq) {
                const bool keep = s.first_seen[i] == s.base + i;
                if (table) {
                    const ckhost::Span id = ckhost::record_id(text, h);
                    if (keep) { kept_slot[s.base + i] = ids.size(); ids.emplace_back((const char*)text + id.off, id.len); }
                    else {                                                     // src/uniq.rs:63-70
                        line.clear();
                        if (!table_header) { line += "id"; line += delim; line += "duplicate_id\n"; table_header = true; }
                        const std::string& fid = ids[kept_slot[s.first_seen[i]]];
                        ckhost::csv_field(line, (const uint8_t*)fid.data(), fid.size(), delim);
                        line += delim;
                        ckhost::csv_field(line, text + id.off, id.len, delim);
                        line += '\n';
                        fwrite(line.data(), 1, line.size(), table);
                    }
                }
                if (!keep) continue;
            }
            // ">" head "\n" sequence "\n"  (src/canonicalize.rs:33-37, src/uniq.rs:50-61)
            const size_t seq_len = want_bytes ? (size_t)(b.offsets[i + 1] - b.offsets[i]) : b.raw[i].len;
            if (mapped_out) { at += h.len + seq_len + 3; continue; }
            iov.push_back({ (void*)&GT, 1 });
            if (h.len) iov.push_back({ (void*)(text + h.off), h.len });
            iov.push_back({ (void*)&NL, 1 });
            if (seq_len) iov.push_back({ (void*)(want_bytes ? s.canon.data() + b.offsets[i] : text + b.raw[i].off), seq_len });
            iov.push_back({ (void*)&NL, 1 });
        }
        if (!mapped_out) { flush_iov(); P.set(seq, FREE); continue; }
        s.ooff[n] = at;
        if (at == 0) { s.emit_left = 0; P.set(seq, ASSEMBLED); continue; }
        // the chunk's place in the file; the emit workers assemble its text and the last of them writes it
        if (s.obuf.size() < at) s.obuf.resize(at + at / 8);
        file_size += at;
        const uint64_t parts = n < (uint64_t)n_emit * 64 ? 1 : (uint64_t)n_emit;
        s.emit_left = (int)parts;
        {
            std::lock_guard<std::mutex> g(jm);
            for (uint64_t k = 0; k < parts; ++k) jobs.push_back(EmitJob{ seq, n * k / parts, n * (k + 1) / parts });
        }
        jcv.notify_all();
    }
    { std::lock_guard<std::mutex> g(jm); jobs_done = true; }
    jcv.notify_all();
    for (auto& t : emitters) t.join();
    if (file_writer.joinable()) file_writer.join();
    reader.join();
    for (auto& t : parsers) t.join();
    gpu.join();
    if (gpu2.joinable()) gpu2.join();
    if (getenv("CIRCKIT_CLI_TIMING"))
        fprintf(stderr, "main() to here %.3f s;  HIP / ctx start-up %.3f s, pipeline %.3f s;  busy: read %.3f s  parse+pack %.3f (sum over %d threads)  gpu %.3f  write/layout %.3f  emit %.3f (sum over %d threads)  file write %.3f\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count(),
                init_s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() - init_s,
                g_read.s(), g_parse.s(), n_parsers, g_gpu.s(), g_write.s(), g_emit.s(), n_emit, g_pwrite.s());
    const auto t_close = std::chrono::steady_clock::now();
    close_output(out);
    if (table) fclose(table);
    const auto t_closed = std::chrono::steady_clock::now();
    close_input(in);
    if (getenv("CIRCKIT_CLI_TIMING"))
        fprintf(stderr, "    close output %.3f s, unmap input %.3f s\n", std::chrono::duration<double>(t_closed - t_close).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_closed).count());
    // Everything is written and closed.  Tearing down 768 MB of pinned buffers, the input mapping and the HIP runtime
    // takes another 0.3-0.6 s that produces nothing: leave it to the kernel (CIRCKIT_CLI_CLEAN_EXIT=1 keeps the orderly
    // teardown, e.g. under a leak checker).
    if (getenv("CIRCKIT_CLI_TIMING") && getenv("CIRCKIT_CLI_TIMING")[0] == '2') {      // what the teardown is made of
        auto lap = [t = std::chrono::steady_clock::now()](const char* what) mutable {
            const auto n = std::chrono::steady_clock::now();
            fprintf(stderr, "    teardown: %s %.3f s\n", what, std::chrono::duration<double>(n - t).count());
            t = n;
        };
        if (in.map) { munmap((void*)in.map, in.map_len); lap("unmap input"); }
        for (Slot& sl : P.slot) { sl.obuf = std::vector<uint8_t>(); }
        lap("free assembled chunks");
        for (Slot& sl : P.slot) { sl.batch.bytes.~ByteBuf(); new (&sl.batch.bytes) ckhost::ByteBuf(); sl.canon.~ByteBuf(); new (&sl.canon) ckhost::ByteBuf(); }
        lap("free pinned buffers");
        circkit_ctx_destroy(ctx);
        if (ctx2) circkit_ctx_destroy(ctx2);
        lap("ctx destroy");
        fflush(nullptr); report_done(); _exit(0);
    }
    if (!getenv("CIRCKIT_CLI_CLEAN_EXIT")) { fflush(nullptr); report_done(); _exit(0); }
    for (Slot& sl : P.slot) { sl.batch.bytes.~ByteBuf(); new (&sl.batch.bytes) ckhost::ByteBuf(); sl.canon.~ByteBuf(); new (&sl.canon) ckhost::ByteBuf(); }   // pinned memory goes before the ctx
    circkit_ctx_destroy(ctx);
    if (ctx2) circkit_ctx_destroy(ctx2);
    fflush(nullptr); report_done();
    return 0;
}
