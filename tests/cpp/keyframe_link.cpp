// C++ caller of the image/tracker boundary through mcptam_hip::KeyFrame (include/mcptam_hip/KeyFrame.hpp).
// `--link-only`: every member is instantiated and every C-ABI symbol it uses must resolve (CPU container; on a box without a
// gfx950 device the constructor throws, which is the documented behaviour -- there is no CPU fallback).
#include <cstdio>
#include <cstring>
#include <vector>
#include "mcptam_hip/KeyFrame.hpp"

using mcptam_hip::KeyFrame;

// never called in --link-only mode, but fully compiled and linked
static int exercise(int w, int h) {
  std::vector<uint8_t> img((size_t)w*h);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) img[(size_t)y*w + x] = (uint8_t)((x*7 + y*13 + ((x/16 + y/16) & 1)*90) & 255);
  KeyFrame a(w, h), b(w, h);
  a.MakeKeyFrame_Lite(img.data(), w); b.MakeKeyFrame_Lite(img.data(), w);
  a.MakeKeyFrame_Rest();
  mcptam_hip::Level L = a.GetLevel(0);
  if (L.w != w || L.h != h || (int)L.vCornerRowLUT.size() != h) return 1;
  a.MakeSBI(); b.MakeSBI();
  auto al = a.IteratePosRelToTarget(b, 4);
  auto sc = a.ScoreKFs({ &b });
  std::vector<mcp_int2> src(L.vCorners.begin(), L.vCorners.begin() + (L.vCorners.size() > 4 ? 4 : L.vCorners.size()));
  auto pm = a.FindPatches(b, 0, src, src, 8);
  mcp_camera cam; std::memset(&cam, 0, sizeof cam);
  auto R = KeyFrame::SE3fromSE2(al.first, cam, cam);
  const double T[12] = {1,0,0, 0,1,0, 0,0,1, 0,0,0};
  std::vector<mcp_td_in> td;
  auto out = a.SearchForPoints(cam, T, T, td, 10, 8);
  std::vector<double> w6;
  auto mu = mcptam_hip::CalcPoseUpdate({}, {}, {}, {}, {}, -1.0, &w6);
  std::vector<mcp_pose_point> pts; double bfw[12]; std::memcpy(bfw, T, sizeof T);
  auto mu2 = mcptam_hip::TrackMapPoseIterations(pts, { cam }, std::vector<double>(T, T + 12), bfw, { 1 }, { -1.0 });
  a.IteratePosRelToLast();
  return (int)(sc.second.size() + pm.size() + out.size()) + (R[0] + mu.first[0] + mu2[0] > 1e300) + a.NumPrev();
}

int main(int argc, char** argv) {
  if (argc > 1 && std::strcmp(argv[1], "--link-only") == 0) {
    std::printf("linked: %d gfx950 device(s) visible\n", mcp_device_count());
    return (void*)&exercise ? 0 : 1;
  }
  try { return exercise(320, 240) >= 0 ? 0 : 1; }
  catch (const std::exception& e) { std::printf("failed: %s\n", e.what()); return 2; }
}
