// reader.cpp -- the text reader of gadf_add_dataset(path) (host code, no GPU needed).
//
// The reference reads a data file with Fortran list-directed input, twice: `read(u, *, iostat=stat) a` per record to count the
// records that begin with a number (gadfit.F90:212-215), then `read(u, *) a, b [, c]` per record to take the first two or three
// numbers of each (gadfit.F90:422-437); records that do not begin with a number -- headers, comments -- are skipped, blank records
// are transparent.  At 1e6 lines that costs flang 2.5 s, at the 1e7 points of the headline workload half a minute, a thousand
// times the fit.  Here the file is mapped once, cut into pieces at line ends, and every piece is parsed on a thread of its own
// (std::from_chars, Fortran's D / Q exponent letters and r*c repeat counts included): 1e6 lines in a few tens of milliseconds.
//
// What a record must look like: values separated by blanks, tabs or ONE comma; the first value decides -- if it is not a number
// the record is skipped, as in the reference; a record that begins with a number but holds fewer values than asked for is an
// error here (the reference's list-directed read would silently continue on the next record and mis-align the columns).
#include "context.h"

#include <charconv>
#include <cstring>
#include <fcntl.h>
#include <memory>
#include <system_error>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

struct gfh_columns {
  int n_columns = 0;
  std::vector<std::vector<double>> piece_x, piece_y, piece_w;      // per piece of the file, in file order
  int64_t n = 0;
};

namespace {

inline bool is_sep(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// one numeric token [p, e): Fortran real syntax -- optional sign, digits with an optional point, optional exponent introduced
// by E, D or Q (either case) or by its sign alone ("1.5+3", "9.6239+195": F2018 13.7.2.3.2; what Fortran's own E / ES edit
// descriptors WRITE once the exponent has three digits).  Returns false unless the whole token is a number.
bool parse_real(const char* p, const char* e, double* out) {
  if (p < e && *p == '+') { p++; if (p < e && (*p == '+' || *p == '-')) return false; }   // from_chars takes no leading '+' ("+-1" is no number)
  if (p >= e) return false;
  char buf[256];
  const size_t len = (size_t)(e - p);
  const char* q = p; const char* qe = e;
  bool letter = false;
  size_t bare = 0;                 // position of a sign that follows the significand directly: an exponent without its letter
  for (size_t k = 0; k < len; k++) {
    const char c = p[k];
    if (c == 'd' || c == 'D' || c == 'q' || c == 'Q') letter = true;
    if (c == 'e' || c == 'E') bare = len;      // (a lettered exponent: its sign is from_chars' business)
    if ((c == '+' || c == '-') && k > 0 && !bare && ((p[k - 1] >= '0' && p[k - 1] <= '9') || p[k - 1] == '.')) bare = k;
  }
  if (bare >= len) bare = 0;
  if (letter || bare) {
    if (len + 1 >= sizeof buf) return false;
    size_t o = 0;
    for (size_t k = 0; k < len; k++) {
      const char c = p[k];
      if (bare && k == bare && !letter) buf[o++] = 'e';
      buf[o++] = (c == 'd' || c == 'D' || c == 'q' || c == 'Q') ? 'e' : c;
    }
    q = buf; qe = buf + o;
  }
  // (an exponent written "E+5": from_chars accepts the sign there)
  double v = 0.0;
  auto r = std::from_chars(q, qe, v, std::chars_format::general);
  if (r.ec != std::errc() || r.ptr != qe) {
    // out-of-range values (1e400) are an error for from_chars; the reference would signal overflow too: not a number here
    return false;
  }
  *out = v;
  return true;
}

struct Piece {
  std::vector<double> x, y, w;
  std::string err;
  int64_t err_line = -1;        // line within the piece (0-based) of the first malformed record
  std::string failure;          // an exception met on the piece's thread (out of memory): reported by the caller
};

// the mapping of a data file, released on every way out
struct Mapping {
  void* p = MAP_FAILED; size_t bytes = 0;
  ~Mapping() { if (p != MAP_FAILED) munmap(p, bytes); }
};
// threads that are joined on every way out (a std::thread destroyed while joinable terminates the process)
struct Joiner {
  std::vector<std::thread> th;
  void join() { for (auto& t : th) if (t.joinable()) t.join(); }
  ~Joiner() { join(); }
};

// the records of [b, e): b is the start of a line, e the end of the mapping or the start of the next piece
void parse_records(const char* b, const char* e, int ncol, Piece* out);
void parse_piece(const char* b, const char* e, int ncol, Piece* out) {
  try { parse_records(b, e, ncol, out); }
  catch (const std::exception& ex) { out->failure = ex.what(); }       // (an exception leaving a thread's function terminates the process)
  catch (...) { out->failure = "unknown exception"; }
}
void parse_records(const char* b, const char* e, int ncol, Piece* out) {
  const char* p = b;
  int64_t line = 0;
  double v[3];
  while (p < e) {
    const char* le = static_cast<const char*>(memchr(p, '\n', (size_t)(e - p)));
    if (!le) le = e;
    const char* q = p;
    int got = 0;
    bool skip = false, bad = false;
    while (q < le && got < ncol) {
      while (q < le && is_sep(*q)) q++;
      if (q >= le) break;
      if (*q == '/') break;                              // a slash ends list-directed input
      if (*q == ',') {                                   // a comma with nothing before it: a null value
        if (got == 0) { skip = true; break; }
        bad = true; break;
      }
      const char* t = q;
      while (q < le && !is_sep(*q) && *q != ',') q++;
      // r*c: r copies of c
      const char* star = static_cast<const char*>(memchr(t, '*', (size_t)(q - t)));
      int reps = 1; const char* vb = t;
      if (star) {
        int r = 0;
        auto rr = std::from_chars(t, star, r);
        if (rr.ec != std::errc() || rr.ptr != star || r < 1) { if (got == 0) skip = true; else bad = true; break; }
        reps = r; vb = star + 1;
      }
      double val;
      if (!parse_real(vb, q, &val)) { if (got == 0) skip = true; else bad = true; break; }
      for (int k = 0; k < reps && got < ncol; k++) v[got++] = val;
      while (q < le && is_sep(*q)) q++;
      if (q < le && *q == ',') q++;                      // the one comma that may separate two values
    }
    if (!skip && !bad && got > 0 && got < ncol) bad = true;            // began with a number, too few values
    if (bad && out->err_line < 0) {
      out->err_line = line;
      out->err = std::string(p, (size_t)std::min<ptrdiff_t>(le - p, 80));
    }
    if (!skip && !bad && got == ncol) {
      out->x.push_back(v[0]); out->y.push_back(v[1]);
      if (ncol == 3) out->w.push_back(v[2]);
    }
    p = le < e ? le + 1 : e;
    line++;
  }
}

}  // namespace

extern "C" {

int gfh_read_columns(const char* path, int n_columns, gfh_columns** out, int64_t* n_points) try {
  if (!path || !out || !n_points || (n_columns != 2 && n_columns != 3)) { gfh::set_global_error("gfh_read_columns: bad arguments"); return 1; }
  *out = nullptr; *n_points = 0;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { gfh::set_global_error(std::string("Cannot open ") + path); return 1; }
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); gfh::set_global_error(std::string("Cannot stat ") + path); return 1; }
  size_t size = (size_t)st.st_size;
  std::unique_ptr<gfh_columns> cols;
  try { cols.reset(new gfh_columns()); } catch (...) { close(fd); throw; }
  cols->n_columns = n_columns;
  // a regular file is mapped; anything else (a pipe, a character device) is read to its end first
  std::string slurped;
  Mapping map;
  const char* base = nullptr;
  if (S_ISREG(st.st_mode) && size > 0) {
    map.p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map.p != MAP_FAILED) { map.bytes = size; base = static_cast<const char*>(map.p); }
  }
  if (!base) {
    try {
      char buf[1 << 16];
      for (;;) {
        const ssize_t got = read(fd, buf, sizeof buf);
        if (got < 0) { close(fd); gfh::set_global_error(std::string("Cannot read ") + path); return 1; }
        if (got == 0) break;
        slurped.append(buf, (size_t)got);
      }
    } catch (...) { close(fd); throw; }
    size = slurped.size(); base = slurped.data();
  }
  close(fd);
  if (size == 0) { *out = cols.release(); return 0; }
  // pieces of >= 4 MB, at most 16 (or what the machine has), each beginning at the start of a line
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  size_t n_pieces = std::min<size_t>(std::min<unsigned>(hw, 16u), std::max<size_t>(1, size / ((size_t)4 << 20)));
  if (const char* e = getenv("GADFIT_HIP_READ_THREADS")) { const int v = atoi(e); if (v >= 1) n_pieces = std::min<size_t>((size_t)v, std::max<size_t>(1, size / 64)); }
  std::vector<const char*> cut(n_pieces + 1);
  cut[0] = base; cut[n_pieces] = base + size;
  for (size_t k = 1; k < n_pieces; k++) {
    const char* p = base + size / n_pieces * k;
    const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(base + size - p)));
    cut[k] = nl ? nl + 1 : base + size;
  }
  for (size_t k = 1; k <= n_pieces; k++) if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
  std::vector<Piece> pieces(n_pieces);
  {
    Joiner workers;
    workers.th.reserve(n_pieces);
    for (size_t k = 1; k < n_pieces; k++) {
      // (a thread that cannot be started: its piece is parsed here instead)
      try { workers.th.emplace_back(parse_piece, cut[k], cut[k + 1], n_columns, &pieces[k]); }
      catch (const std::system_error&) { parse_piece(cut[k], cut[k + 1], n_columns, &pieces[k]); }
    }
    parse_piece(cut[0], cut[1], n_columns, &pieces[0]);
    workers.join();
  }
  for (size_t k = 0; k < n_pieces; k++)
    if (!pieces[k].failure.empty()) { gfh::set_global_error(std::string(path) + ": " + pieces[k].failure); return 1; }
  // the first malformed record in file order
  for (size_t k = 0; k < n_pieces; k++) {
    if (pieces[k].err_line >= 0) {
      int64_t ln = 1;
      for (const char* p = base; p < cut[k]; p++) if (*p == '\n') ln++;
      gfh::set_global_error(std::string(path) + ", line " + std::to_string(ln + pieces[k].err_line) + ": fewer than " + std::to_string(n_columns) +
                            " numbers in a record that begins with one: '" + pieces[k].err + "'");
      return 1;
    }
  }
  cols->piece_x.resize(n_pieces); cols->piece_y.resize(n_pieces); cols->piece_w.resize(n_pieces);
  for (size_t k = 0; k < n_pieces; k++) {
    cols->n += (int64_t)pieces[k].x.size();
    cols->piece_x[k] = std::move(pieces[k].x); cols->piece_y[k] = std::move(pieces[k].y); cols->piece_w[k] = std::move(pieces[k].w);
  }
  *n_points = cols->n; *out = cols.release();
  return 0;
} catch (const std::exception& e) { gfh::set_global_error(std::string("gfh_read_columns: ") + e.what()); return 1; }

int gfh_take_columns(gfh_columns* cols, double* x, double* y, double* w) {
  if (!cols) { gfh::set_global_error("gfh_take_columns: null handle"); return 1; }
  int64_t off = 0;
  for (size_t k = 0; k < cols->piece_x.size(); k++) {
    const size_t n = cols->piece_x[k].size();
    if (n) {
      if (x) memcpy(x + off, cols->piece_x[k].data(), sizeof(double) * n);
      if (y) memcpy(y + off, cols->piece_y[k].data(), sizeof(double) * n);
      if (w && cols->n_columns == 3) memcpy(w + off, cols->piece_w[k].data(), sizeof(double) * n);
    }
    off += (int64_t)n;
  }
  delete cols;
  return 0;
}

void gfh_free_columns(gfh_columns* cols) { delete cols; }

}  // extern "C"
