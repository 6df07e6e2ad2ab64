// TEST INFRASTRUCTURE ONLY — part of oracle/_ref/libref.so (oracle/Makefile.ref).
// Implements the reference's image-reader interface (src/dso/IOWrapper/ImageRW.h; its OpenCV back end ImageRW_OpenCV.cpp cannot be built here) over an in-memory
// registry: a test registers a buffer under a name (ref_register_image16 / ref_register_image8) and the reference's own code — PhotometricUndistorter's constructor reading
// its vignette image, util/Undistort.cpp:120-121 — "reads" it from there.  Writers are no-ops like in ImageRW_dummy.cpp.
#include <map>
#include <string>
#include <vector>
#include <cstring>
#include "IOWrapper/ImageRW.h"

namespace {
struct Stored { int w, h, bits; std::vector<unsigned char> bytes; };
std::map<std::string, Stored>& registry() { static std::map<std::string, Stored> r; return r; }
}

extern "C" {
void ref_register_image16(const char* name, const unsigned short* data, int w, int h) {
  Stored s; s.w = w; s.h = h; s.bits = 16; s.bytes.resize((size_t)w * h * 2); memcpy(s.bytes.data(), data, s.bytes.size()); registry()[name] = s;
}
void ref_register_image8(const char* name, const unsigned char* data, int w, int h) {
  Stored s; s.w = w; s.h = h; s.bits = 8; s.bytes.resize((size_t)w * h); memcpy(s.bytes.data(), data, s.bytes.size()); registry()[name] = s;
}
void ref_unregister_images() { registry().clear(); }
}

namespace dso {
namespace IOWrap {

MinimalImageB* readImageBW_8U(std::string filename) {
  auto it = registry().find(filename);
  if (it == registry().end() || it->second.bits != 8) return 0;
  MinimalImageB* img = new MinimalImageB(it->second.w, it->second.h);
  memcpy(img->data, it->second.bytes.data(), it->second.bytes.size());
  return img;
}
MinimalImageB3* readImageRGB_8U(std::string) { return 0; }
MinimalImage<unsigned short>* readImageBW_16U(std::string filename) {
  auto it = registry().find(filename);
  if (it == registry().end() || it->second.bits != 16) return 0;
  MinimalImage<unsigned short>* img = new MinimalImage<unsigned short>(it->second.w, it->second.h);
  memcpy(img->data, it->second.bytes.data(), it->second.bytes.size());
  return img;
}
MinimalImageB* readStreamBW_8U(char*, int) { return 0; }
void writeImage(std::string, MinimalImageB*) {}
void writeImage(std::string, MinimalImageB3*) {}
void writeImage(std::string, MinimalImageF*) {}
void writeImage(std::string, MinimalImageF3*) {}

}  // namespace IOWrap
}  // namespace dso
