// formats.hip — "next" row (f)3 of SURVEY.md section 8: the on-disk formats either side of the path (host code only;
// nothing here touches the GPU).
//   * KITTI Velodyne .bin: float32 x, y, z, intensity records; the demo's loader reads at most 1 000 000 floats =
//     250 000 points (examples/run_global_registration.cpp:377-402).
//   * matched-pair cache "%06d_to_%06d.pcd" (FPFHManager::saveFeaturePair / loadFeaturePair,
//     include/fpfh_manager.hpp:179-232): source key points then target key points in ONE cloud, written with
//     pcl::io::savePCDFile's defaults (PCD v0.7, DATA ascii, 8 significant digits — PCL 1.8 PCDWriter::writeASCII;
//     PCL is a third-party dependency absent from the reference tree, its published file format is restated here).
//     The reader takes the three DATA kinds PCL writes (ascii, binary, binary_compressed = LZF over field-major data).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/quatro_hip.h"

namespace {

struct PcdField {
  std::string name;
  int size = 4, count = 1;
  char type = 'F';
  int offset = 0;  // byte offset inside one binary record
};

struct PcdHeader {
  std::vector<PcdField> fields;
  long long width = 0, height = 1, points = -1;
  std::string data;
  int record_bytes = 0;
};

bool read_line(FILE* f, std::string& line) {
  line.clear();
  int c;
  while ((c = fgetc(f)) != EOF) {
    if (c == '\n') return true;
    if (c != '\r') line.push_back((char)c);
  }
  return !line.empty();
}

std::vector<std::string> split_ws(const std::string& s) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) ++i;
    size_t j = i;
    while (j < s.size() && s[j] != ' ' && s[j] != '\t') ++j;
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j;
  }
  return out;
}

// header up to and including the DATA line; leaves the stream at the first payload byte
bool read_pcd_header(FILE* f, PcdHeader& h) {
  std::string line;
  while (read_line(f, line)) {
    if (line.empty() || line[0] == '#') continue;
    const std::vector<std::string> w = split_ws(line);
    if (w.empty()) continue;
    const std::string& key = w[0];
    if (key == "VERSION" || key == "VIEWPOINT") continue;
    if (key == "FIELDS" || key == "COLUMNS") {
      h.fields.resize(w.size() - 1);
      for (size_t i = 1; i < w.size(); ++i) h.fields[i - 1].name = w[i];
    } else if (key == "SIZE" || key == "TYPE" || key == "COUNT") {
      if (w.size() - 1 != h.fields.size()) return false;
      for (size_t i = 1; i < w.size(); ++i) {
        if (key == "SIZE") h.fields[i - 1].size = atoi(w[i].c_str());
        if (key == "TYPE") h.fields[i - 1].type = w[i][0];
        if (key == "COUNT") h.fields[i - 1].count = atoi(w[i].c_str());
      }
    } else if (key == "WIDTH" && w.size() > 1) {
      h.width = atoll(w[1].c_str());
    } else if (key == "HEIGHT" && w.size() > 1) {
      h.height = atoll(w[1].c_str());
    } else if (key == "POINTS" && w.size() > 1) {
      h.points = atoll(w[1].c_str());
    } else if (key == "DATA" && w.size() > 1) {
      h.data = w[1];
      if (h.points < 0) h.points = h.width * h.height;
      long long off = 0;  // sizes come from the file: bound them before any arithmetic or allocation depends on them
      for (PcdField& fl : h.fields) {
        if ((fl.size != 1 && fl.size != 2 && fl.size != 4 && fl.size != 8) || fl.count < 1 || fl.count > 65536) return false;
        fl.offset = (int)off;
        off += (long long)fl.size * fl.count;
        if (off > (1 << 20)) return false;
      }
      h.record_bytes = (int)off;
      return !h.fields.empty() && h.points >= 0 && h.width >= 0 && h.height >= 0;
    }
  }
  return false;
}

float load_scalar(const unsigned char* p, const PcdField& f) {
  if (f.type == 'F' && f.size == 4) {
    float v;
    memcpy(&v, p, 4);
    return v;
  }
  if (f.type == 'F' && f.size == 8) {
    double v;
    memcpy(&v, p, 8);
    return (float)v;
  }
  long long s = 0;
  unsigned long long u = 0;
  memcpy(&u, p, (size_t)(f.size > 8 ? 8 : f.size));
  if (f.type == 'U') return (float)u;
  if (f.size == 1) s = (int8_t)u;
  else if (f.size == 2) s = (int16_t)u;
  else if (f.size == 4) s = (int32_t)u;
  else s = (long long)u;
  return (float)s;
}

// liblzf stream: control byte c < 32 -> c + 1 literals; otherwise a back reference of length (c >> 5) + 2 (an extra
// length byte follows when c >> 5 == 7) at distance ((c & 31) << 8 | next byte) + 1
bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    const unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t run = ctrl + 1;
      if (ip + run > in_len || op + run > out_len) return false;
      memcpy(out + op, in + ip, run);
      ip += run;
      op += run;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) {
        if (ip >= in_len) return false;
        len += in[ip++];
      }
      if (ip >= in_len) return false;
      const size_t dist = (((size_t)(ctrl & 31)) << 8 | in[ip++]) + 1;
      len += 2;
      if (dist > op || op + len > out_len) return false;
      for (size_t i = 0; i < len; ++i, ++op) out[op] = out[op - dist];  // may overlap: byte by byte
    }
  }
  return op == out_len;
}

}  // namespace

extern "C" {

int qtr_read_kitti_bin(const char* path, float* xyzi, int max_points, int* n_points) {
  if (!path || !n_points || max_points < 0 || (max_points > 0 && !xyzi)) return QTR_ERR_BAD_ARG;
  *n_points = 0;
  FILE* f = fopen(path, "rb");
  if (!f) return QTR_ERR_IO;  // the demo prints "error: failed to load" and returns nullptr (:379-382)
  const size_t got = max_points > 0 ? fread(xyzi, sizeof(float), (size_t)max_points * 4, f) : 0;
  fclose(f);
  *n_points = (int)(got / 4);  // a trailing partial record is dropped, as the demo's integer division does (:385-386)
  return QTR_OK;
}

int qtr_write_pcd_xyz(const char* path, const float* xyz4, int n, int binary) {
  if (!path || n < 0 || (n > 0 && !xyz4)) return QTR_ERR_BAD_ARG;
  FILE* f = fopen(path, "wb");
  if (!f) return QTR_ERR_IO;
  fprintf(f,
          "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
          "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n",
          n, n, binary ? "binary" : "ascii");
  bool ok = true;
  for (int i = 0; i < n && ok; ++i) {
    const float* p = xyz4 + 4 * (size_t)i;
    if (binary) {
      ok = fwrite(p, 4, 3, f) == 3;
    } else {
      for (int c = 0; c < 3; ++c) {
        if (std::isnan(p[c]))
          ok = ok && fputs("nan", f) >= 0;
        else
          ok = ok && fprintf(f, "%.8g", (double)p[c]) > 0;  // stream precision 8, default float format
        ok = ok && fputc(c < 2 ? ' ' : '\n', f) != EOF;
      }
    }
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? QTR_OK : QTR_ERR_IO;
}

int qtr_read_pcd_xyz(const char* path, float* xyz4, int cap, int* n_points) {
  if (!path || !n_points || cap < 0 || (cap > 0 && !xyz4)) return QTR_ERR_BAD_ARG;
  *n_points = 0;
  FILE* f = fopen(path, "rb");
  if (!f) return QTR_ERR_IO;
  PcdHeader h;
  if (!read_pcd_header(f, h)) {
    fclose(f);
    return QTR_ERR_IO;
  }
  int fi[3] = {-1, -1, -1};
  for (size_t i = 0; i < h.fields.size(); ++i)
    for (int c = 0; c < 3; ++c)
      if (h.fields[i].name == (c == 0 ? "x" : c == 1 ? "y" : "z")) fi[c] = (int)i;
  if (fi[0] < 0 || fi[1] < 0 || fi[2] < 0 || h.points > 0x7fffffffLL) {
    fclose(f);
    return QTR_ERR_IO;
  }
  const int n = (int)h.points;
  *n_points = n;
  if (n > cap) {
    fclose(f);
    return QTR_ERR_CAPACITY;
  }
  int rc = QTR_OK;
  if (h.data == "ascii") {
    std::string line;
    int tokens_per_point = 0;
    for (const PcdField& fl : h.fields) tokens_per_point += fl.count;
    std::vector<int> tok_of(3);
    for (int c = 0; c < 3; ++c) {
      int t = 0;
      for (int i = 0; i < fi[c]; ++i) t += h.fields[(size_t)i].count;
      tok_of[(size_t)c] = t;
    }
    for (int i = 0; i < n; ++i) {
      do {
        if (!read_line(f, line)) {
          rc = QTR_ERR_IO;
          break;
        }
      } while (line.empty());
      if (rc != QTR_OK) break;
      const std::vector<std::string> w = split_ws(line);
      if ((int)w.size() < tokens_per_point) {
        rc = QTR_ERR_IO;
        break;
      }
      float* p = xyz4 + 4 * (size_t)i;
      for (int c = 0; c < 3; ++c) p[c] = strtof(w[(size_t)tok_of[(size_t)c]].c_str(), nullptr);  // "nan" parses as NaN
      p[3] = 0.f;
    }
  } else if (h.data == "binary" || h.data == "binary_compressed") {
    const size_t total = (size_t)n * (size_t)h.record_bytes;
    const bool soa = h.data == "binary_compressed";
    // what is left of the file bounds what a truthful header can announce (no allocation on a header's word alone)
    const long here = ftell(f);
    long left = 0;
    if (here >= 0 && fseek(f, 0, SEEK_END) == 0) {
      left = ftell(f) - here;
      (void)fseek(f, here, SEEK_SET);
    }
    if (left < 0 || (!soa && (size_t)left < total) || (soa && (left < 8 || total > ((size_t)1 << 31)))) {
      fclose(f);
      return QTR_ERR_IO;
    }
    std::vector<unsigned char> buf(total ? total : 1);
    if (soa) {
      uint32_t sizes[2] = {0, 0};
      if (fread(sizes, 4, 2, f) != 2 || sizes[1] != total || (size_t)sizes[0] > (size_t)left - 8) {
        rc = QTR_ERR_IO;
      } else {
        std::vector<unsigned char> comp(sizes[0] ? sizes[0] : 1);
        if (fread(comp.data(), 1, sizes[0], f) != sizes[0] || !lzf_decompress(comp.data(), sizes[0], buf.data(), total))
          rc = QTR_ERR_IO;
      }
    } else if (fread(buf.data(), 1, total, f) != total) {
      rc = QTR_ERR_IO;
    }
    if (rc == QTR_OK) {
      for (int i = 0; i < n; ++i) {
        float* p = xyz4 + 4 * (size_t)i;
        for (int c = 0; c < 3; ++c) {
          const PcdField& fl = h.fields[(size_t)fi[c]];
          const unsigned char* src = soa ? buf.data() + (size_t)fl.offset * (size_t)n + (size_t)i * (size_t)(fl.size * fl.count)
                                         : buf.data() + (size_t)i * (size_t)h.record_bytes + (size_t)fl.offset;
          p[c] = load_scalar(src, fl);
        }
        p[3] = 0.f;
      }
    }
  } else {
    rc = QTR_ERR_IO;
  }
  fclose(f);
  return rc;
}

}  // extern "C"
