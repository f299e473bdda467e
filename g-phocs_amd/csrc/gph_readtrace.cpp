// Post-run summary of a trace file: block means per column.  Host-only; the counterpart of the
// reference's stand-alone `readTrace` tool (src/readTrace.c:41-291), exported through the C ABI as
// gph_read_trace() and built as the `readTrace` executable next to G-PhoCS-hip.
//
// What is kept from readTrace.c: the first column (sample index) is dropped (:160), values are parsed
// as single-precision floats (`%f` into a float, :225) and summed in long double (:227), a block mean is
// the sum divided by the block size, the default block is the whole file (:137-139), `-d N` skips the
// first N samples and fails when N >= the number of samples (:140-143), a trailing partial block is
// averaged over its own length (:253-264), a last line without a newline is not counted (:217-218),
// means are printed as "%.6f" followed by four blanks, left-justified in a column as wide as the widest
// entry of that column (:229-233, :269-287), titles above the first block only, at most 90 columns per
// printed row (:279), one empty line at the end (:291).
// What is not: fixed 4096-byte line and row buffers (any length is accepted here), and reading
// uninitialised memory (upstream never zeroes its sums / widths and stores a trailing block's means only
// where they widen the column; here the sums and widths start at zero and every mean is stored).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/gphocs_hip.h"

namespace {

struct Out {
  std::string s;
  void put(const char *p) { s += p; }
};

void split(const std::string &line, std::vector<std::string> &tok)
{
  tok.clear();
  size_t i = 0, n = line.size();
  while (i < n) {
    while (i < n && (line[i] == ' ' || line[i] == '\t')) i++;
    size_t j = i;
    while (j < n && line[j] != ' ' && line[j] != '\t') j++;
    if (j > i) tok.push_back(line.substr(i, j - i));
    i = j;
  }
}

// fgets semantics without the length limit: returns false at end of file; `eof_hit` = the line ended
// at end of file rather than at a newline
bool next_line(FILE *f, std::string &line, bool &eof_hit)
{
  line.clear();
  eof_hit = false;
  int c;
  while ((c = fgetc(f)) != EOF) {
    line.push_back((char)c);
    if (c == '\n') return true;
  }
  eof_hit = true;
  return !line.empty();
}

std::string pad(const std::string &s, int w)
{
  std::string r = s;
  if ((int)r.size() < w) r.append((size_t)(w - (int)r.size()), ' ');
  return r;
}

int summarize(const char *path, int blockSize, int discard, Out &out, std::string &err)
{
  FILE *f = fopen(path, "r");
  if (!f) { err = std::string("Could not find trace file '") + path + "' specified.\n"; return 1; }
  std::string line, header;
  bool eofh;
  if (!next_line(f, header, eofh)) { fclose(f); err = "Unable to discard header of the trace file\n"; return -1; }
  int numLines = 0;
  while (next_line(f, line, eofh)) numLines++;
  if (blockSize < 0) blockSize = numLines;
  if (discard >= numLines) {
    char b[160];
    snprintf(b, sizeof b, "%d lines specified to discard, but trace file contains only %d lines.\n", discard, numLines);
    err = b;
    fclose(f);
    return 1;
  }
  if (blockSize == 0) { fclose(f); err = "Block size must not be zero.\n"; return 1; }
  fseek(f, 0, SEEK_SET);
  next_line(f, header, eofh);
  if (!header.empty() && header.back() == '\n') header.back() = ' ';
  std::vector<std::string> names;
  split(header, names);
  if (names.empty()) { fclose(f); err = "Unable to get the first line of the trace file.\n"; return -1; }
  names.erase(names.begin());
  const int numCols = (int)names.size();
  std::vector<int> width(numCols, 0);
  std::vector<long double> sums(numCols, 0.0L);
  std::vector<std::vector<double>> data;
  for (int i = 0; i < discard; i++)
    if (!next_line(f, line, eofh)) { fclose(f); err = "Unable to discard lines.\n"; return -1; }
  int count = 0;
  std::vector<std::string> tok;
  char tmp[128];
  auto close_block = [&](int len) {
    std::vector<double> row(numCols);
    for (int c = 0; c < numCols; c++) {
      snprintf(tmp, sizeof tmp, "%.6Lf    ", sums[c] / len);
      if ((int)strlen(tmp) > width[c]) width[c] = (int)strlen(tmp);
      row[c] = (double)(sums[c] / len);
      sums[c] = 0.0L;
    }
    data.push_back(row);
  };
  while (next_line(f, line, eofh)) {
    if (eofh) break;                       /* a last line without its newline is not a sample */
    if (!line.empty() && line.back() == '\n') line.pop_back();
    split(line, tok);
    if ((int)tok.size() < numCols + 1) {
      fclose(f);
      err = "A trace line has fewer columns than the header.\n";
      return -1;
    }
    count++;
    for (int c = 0; c < numCols; c++) {
      float v = strtof(tok[c + 1].c_str(), nullptr);
      sums[c] = sums[c] + v;
    }
    if (count == blockSize) { close_block(blockSize); count = 0; }
  }
  fclose(f);
  if (count > 0) close_block(count);
  std::string valueLine, titleLine;
  for (size_t i = 0; i < data.size(); i++) {
    for (int c = 0; c < numCols; c++) {
      snprintf(tmp, sizeof tmp, "%.6f    ", data[i][c]);
      valueLine += pad(tmp, width[c]);
      if (i == 0) titleLine += pad(names[c], width[c]);
      if (((c + 1) % 90) == 0 || c == numCols - 1) {
        if (i == 0) { out.put(titleLine.c_str()); out.put("\n"); }
        out.put(valueLine.c_str());
        out.put("\n");
        titleLine.clear();
        valueLine.clear();
      }
    }
  }
  out.put("\n");
  return 0;
}

}  // namespace

extern "C" int gph_read_trace(const char *trace_file, int block_size, int discard, char *out, size_t out_cap,
                              size_t *out_len, char *err, size_t err_cap)
{
  Out o;
  std::string e;
  int rc = summarize(trace_file, block_size, discard, o, e);
  if (out_len) *out_len = o.s.size();
  if (out && out_cap) {
    size_t n = o.s.size() < out_cap - 1 ? o.s.size() : out_cap - 1;
    memcpy(out, o.s.data(), n);
    out[n] = 0;
  }
  if (err && err_cap) {
    size_t n = e.size() < err_cap - 1 ? e.size() : err_cap - 1;
    memcpy(err, e.data(), n);
    err[n] = 0;
  }
  if (rc == 0 && out && o.s.size() + 1 > out_cap) return 2;   /* output truncated: call again with *out_len + 1 */
  return rc;
}

#ifdef GPH_READTRACE_MAIN
// readTrace <trace-file-name> [-b SIZE] [-d NUMBER] [-h]     (readTrace.c:14-39, 57-102)
static void print_help()
{
  printf("-b, --block-size  SIZE     Blocksize\n");
  printf("-d, --discard  NUMBER      Number of samples from to discard from beginning of file\n");
  printf("-h, --help                 This help page\n");
}
static void print_usage(const char *a0)
{
  printf("Usage: %s <trace-file-name> [options]\n", a0);
  print_help();
}
int main(int argc, char **argv)
{
  int block = -1, discard = 0;
  const char *file = nullptr;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto need = [&](const char *name) -> const char * {
      if (i + 1 >= argc) { fprintf(stderr, "Option -%s requires an argument.\n", name); exit(1); }
      return argv[++i];
    };
    if (a == "-b" || a == "--block-size") block = atoi(need("b"));
    else if (a.rfind("--block-size=", 0) == 0) block = atoi(a.c_str() + 13);
    else if (a.rfind("-b", 0) == 0 && a.size() > 2) block = atoi(a.c_str() + 2);
    else if (a == "-d" || a == "--discard") discard = atoi(need("d"));
    else if (a.rfind("--discard=", 0) == 0) discard = atoi(a.c_str() + 10);
    else if (a.rfind("-d", 0) == 0 && a.size() > 2) discard = atoi(a.c_str() + 2);
    else if (a == "-s" || a == "--sub-sampling" || a == "-t") (void)need("s");   /* accepted and ignored upstream */
    else if (a == "-h" || a == "--help") { print_usage(argv[0]); print_help(); return 0; }
    else if (a.size() > 1 && a[0] == '-') { fprintf(stderr, "Unknown option `%s'.\n", a.c_str()); return 1; }
    else if (!file) file = argv[i];
  }
  if (!file) {
    fprintf(stderr, "Missing trace filename.\n");
    print_usage(argv[0]);
    return 1;
  }
  size_t len = 0;
  char errb[512] = "";
  gph_read_trace(file, block, discard, nullptr, 0, &len, errb, sizeof errb);
  std::vector<char> buf(len + 1);
  int rc = gph_read_trace(file, block, discard, buf.data(), buf.size(), &len, errb, sizeof errb);
  if (rc != 0) {
    fputs(errb, stderr);
    if (rc == 1 && strstr(errb, "Could not find")) print_usage(argv[0]);
    return rc;
  }
  fputs(buf.data(), stdout);
  return 0;
}
#endif
