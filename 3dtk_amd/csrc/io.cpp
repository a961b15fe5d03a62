// On-disk formats either side of the hot path: the uos ASCII scan reader with
// the -m/-M range filter, the .pose reader and the .frames writer.  Host-only, no device.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tdtk_internal.h"

using namespace tdtk;

extern "C" {

// uos ".3d" reader: one "x y z" triple per line, '#' comments, blank lines, and up to 10
// unparsable lines at the top (the optional point-count header) are skipped, like readASCII
// (src/scanio/helper.cc:564-700, 730-880).  Values are parsed with strtod.  The range filter is
// PointFilter::setRange (src/slam6d/pointfilter.cc:162-188): keep x^2+y^2+z^2 < max^2 (max > 0)
// and > min^2 (min > 0).  setRange hands its two doubles to the checkers as TEXT (`stringstream << maxDist`, default
// precision: six significant digits, pointfilter.cc:63-70 -> :162-188), so a range of 123.456789 filters at 123.457: the
// same round trip is made here (%g is what operator<< prints).  -m / -M are ints on slam6D's command line (slam6D.cc:224,241),
// for which the trip is exact up to 999999.  *xyz_out is malloc'ed; release with tdtk_io_free.
static double through_text(double v)
{
  char buf[64];
  std::snprintf(buf, sizeof buf, "%g", v);
  return std::strtod(buf, nullptr);
}

int tdtk_io_read_uos(const char* path, double range_max, double range_min, double** xyz_out, size_t* n_out)
{
  if (!path || !xyz_out || !n_out) { set_error("NULL argument"); return TDTK_EINVAL; }
  FILE* f = std::fopen(path, "r");
  if (!f) { set_error(std::string("cannot open ") + path); return TDTK_EINVAL; }
  std::vector<double> pts;
  char line[4096];
  int header_budget = 10;
  unsigned long linenr = 0;
  const double rmax = through_text(range_max), rmin = through_text(range_min);
  const double max2 = rmax > 0.0 ? rmax * rmax : -1.0;
  const double min2 = rmin > 0.0 ? rmin * rmin : -1.0;
  while (std::fgets(line, sizeof line, f)) {
    ++linenr;
    char* hash = std::strchr(line, '#');
    if (hash) *hash = '\0';
    char* p = line;
    double v[3];
    int got = 0;
    bool bad = false;
    while (got < 4) {
      while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') ++p;
      if (*p == '\0') break;
      char* e;
      const double x = std::strtod(p, &e);
      if (e == p) { bad = true; break; }
      if (got < 3) v[got] = x;
      ++got;
      p = e;
    }
    if (got == 0 && !bad) continue;  // empty / comment line
    if (bad || got != 3) {
      if (header_budget-- > 0 && pts.empty()) continue;  // liberal about garbage at the top
      std::fclose(f);
      set_error(std::string("can't understand line ") + std::to_string(linenr) + " of " + path);
      return TDTK_EINVAL;
    }
    const double r2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (max2 > 0.0 && !(r2 < max2)) continue;
    if (min2 > 0.0 && !(r2 > min2)) continue;
    pts.push_back(v[0]); pts.push_back(v[1]); pts.push_back(v[2]);
  }
  std::fclose(f);
  double* out = (double*)std::malloc(sizeof(double) * (pts.empty() ? 1 : pts.size()));
  if (!out) { set_error("out of memory"); return TDTK_ENOMEM; }
  std::memcpy(out, pts.data(), sizeof(double) * pts.size());
  *xyz_out = out;
  *n_out = pts.size() / 3;
  return TDTK_OK;
}

void tdtk_io_free(void* p) { std::free(p); }

// ".pose": 6 plain doubles, position then Euler angles in degrees -> radians with rad()
// (src/scanio/helper.cc:226-231, include/slam6d/globals.icc:172-175: (2*M_PI*deg)/360)
int tdtk_io_read_pose(const char* path, double rPos[3], double rPosTheta[3])
{
  if (!path || !rPos || !rPosTheta) { set_error("NULL argument"); return TDTK_EINVAL; }
  FILE* f = std::fopen(path, "r");
  if (!f) { set_error(std::string("cannot open ") + path); return TDTK_EINVAL; }
  double v[6];
  int got = 0;
  for (; got < 6; got++)
    if (std::fscanf(f, "%lf", &v[got]) != 1) break;
  std::fclose(f);
  if (got != 6) { set_error(std::string("pose file needs 6 values: ") + path); return TDTK_EINVAL; }
  for (int k = 0; k < 3; k++) {
    rPos[k] = v[k];
    rPosTheta[k] = (2 * M_PI * v[3 + k]) / 360;
  }
  return TDTK_OK;
}

// ".frames": one line per transform event = 16 matrix entries at default ostream precision
// (6 significant digits, operator<< of globals.icc:123-131) followed by the AlgoType
// (0 INVALID, 1 ICP, 2 ICPINACTIVE, 3 LUM, 4 ELCH; include/slam6d/scan.h:126)
// (BasicScan::saveFrames, src/slam6d/basicScan.cc:902-917)
int tdtk_io_write_frames(const char* path, const double* transMats, const int* types, size_t count, int append)
{
  if (!path || (count && (!transMats || !types))) { set_error("NULL argument"); return TDTK_EINVAL; }
  FILE* f = std::fopen(path, append ? "a" : "w");
  if (!f) { set_error(std::string("cannot open ") + path); return TDTK_EINVAL; }
  for (size_t k = 0; k < count; k++) {
    for (int i = 0; i < 16; i++) {
      const double v = transMats[16 * k + i];
      if (std::isnan(v)) { std::fclose(f); set_error("will not write out NAN value"); return TDTK_EINVAL; }
      std::fprintf(f, "%g ", v);
    }
    std::fprintf(f, "%d\n", types[k]);
  }
  std::fclose(f);
  return TDTK_OK;
}

}  // extern "C"
