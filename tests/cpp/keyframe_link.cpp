// C++ caller of the image/tracker boundary through mcptam_hip::KeyFrame (include/mcptam_hip/KeyFrame.hpp).
// `--link-only`: every member is instantiated and every C-ABI symbol it uses must resolve (CPU container; on a box without a
// gfx950 device the constructor throws, which is the documented behaviour -- there is no CPU fallback).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "mcptam_hip/KeyFrame.hpp"

using mcptam_hip::KeyFrame;

// fully compiled and linked in --link-only mode; run on a GPU box by tests/test_cpp_host.py::test_cpp_keyframe_mirror_runs_on_gpu.
// Returns 0 when every member behaves: the same textured frame in two KeyFrames must give identical pyramids and corners, a
// self-alignment of the small blurry images that is the identity, a relocaliser score of 0, MiniPatch matches at zero offset,
// and a pose update of zero from measurements that sit exactly on their projections.
#define CHECK(cond) do { if (!(cond)) { std::printf("check failed, line %d: %s\n", __LINE__, #cond); return 1; } } while (0)
static int exercise(int w, int h) {
  std::vector<uint8_t> img((size_t)w*h);
  unsigned s = 12345;
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
    s = s*1664525u + 1013904223u;
    img[(size_t)y*w + x] = (uint8_t)(((x/12 + y/9) & 1)*120 + ((x*5 + y*3) & 31) + ((s >> 24) & 15) + 40);      // checkerboard + ramp + noise
  }
  KeyFrame a(w, h), b(w, h);
  a.MakeKeyFrame_Lite(img.data(), w); b.MakeKeyFrame_Lite(img.data(), w);
  CHECK(a.NumPrev() == 0);
  a.MakeKeyFrame_Rest(); b.MakeKeyFrame_Rest();
  size_t total_corners = 0;
  for (int l = 0; l < MCP_LEVELS; ++l) {
    mcptam_hip::Level L = a.GetLevel(l), M = b.GetLevel(l);
    CHECK(L.w == (w >> l) && L.h == (h >> l) && (int)L.vCornerRowLUT.size() == L.h);
    CHECK(L.image == M.image && L.vCorners.size() == M.vCorners.size() && L.nFastThresh == M.nFastThresh);
    for (size_t i = 0; i < L.vCorners.size(); ++i) CHECK(L.vCorners[i].x == M.vCorners[i].x && L.vCorners[i].y == M.vCorners[i].y);
    for (size_t i = 1; i < L.vCorners.size(); ++i)        // raster order
      CHECK(L.vCorners[i].y > L.vCorners[i-1].y || (L.vCorners[i].y == L.vCorners[i-1].y && L.vCorners[i].x > L.vCorners[i-1].x));
    for (int y = 1; y < L.h; ++y) CHECK(L.vCornerRowLUT[y] >= L.vCornerRowLUT[y-1]);
    CHECK(L.nFastThresh >= MCP_MIN_FAST_THRESH && L.nFastThresh <= MCP_MAX_FAST_THRESH);
    CHECK(L.vCandidates.size() == L.vCandidateScores.size() && L.vCandidates.size() <= L.vCorners.size());
    total_corners += L.vCorners.size();
  }
  CHECK(total_corners > 100);
  mcptam_hip::Level L = a.GetLevel(0);
  a.MakeSBI(); b.MakeSBI();
  auto al = a.IteratePosRelToTarget(b, 4);
  CHECK(std::fabs(al.first[0] - 1.0) < 1e-9 && std::fabs(al.first[3] - 1.0) < 1e-9 && std::fabs(al.first[4]) < 1e-9 && std::fabs(al.first[5]) < 1e-9 && al.second < 1e-6);
  auto sc = a.ScoreKFs({ &b, nullptr });
  CHECK(sc.first == 0 && sc.second.size() == 2 && sc.second[0] == 0.0);
  std::vector<mcp_int2> src;
  for (size_t i = 0; i < L.vCorners.size() && src.size() < 50; i += 7)
    if (L.vCorners[i].x > 12 && L.vCorners[i].y > 12 && L.vCorners[i].x < w - 12 && L.vCorners[i].y < h - 12) src.push_back(L.vCorners[i]);
  CHECK(src.size() > 10);
  auto pm = a.FindPatches(b, 0, src, src, 8);
  for (size_t i = 0; i < pm.size(); ++i) CHECK(pm[i].found && pm[i].ssd == 0 && pm[i].pos.x == src[i].x && pm[i].pos.y == src[i].y);
  // a second frame: the first one moves into the device-resident history, the candidates get the stability pruning
  a.MakeKeyFrame_Lite(img.data(), w);
  CHECK(a.NumPrev() == 1);
  a.MakeKeyFrame_Rest();
  a.MakeSBI();
  auto last = a.IteratePosRelToLast();
  CHECK(std::fabs(last.first[0] - 1.0) < 1e-9 && std::fabs(last.first[4]) < 1e-9);
  // pose update from perfect measurements: mu = 0
  const int n = 40;
  std::vector<uint8_t> found(n, 1); std::vector<double> fp(2*n), ip(2*n), sn(n, 1.0), J(12*n, 0.0), w6;
  for (int i = 0; i < n; ++i) { fp[2*i] = ip[2*i] = 10.0 + 3*i; fp[2*i+1] = ip[2*i+1] = 20.0 + i; for (int k = 0; k < 6; ++k) { J[12*i + k] = 1.0 + 0.1*k + 0.01*i; J[12*i + 6 + k] = 0.5 - 0.05*k + 0.02*i; } }
  auto mu = mcptam_hip::CalcPoseUpdate(found, fp, ip, sn, J, 1.0, &w6);      // (sigma^2 given: the median of all-zero errors would be 0)
  for (int k = 0; k < 6; ++k) CHECK(mu.first[k] == 0.0);
  for (int i = 0; i < n; ++i) CHECK(w6[i] == 1.0);
  // still instantiated (their numerics are covered through ctypes in tests/test_img_gpu.py): the camera-dependent members
  mcp_camera cam; std::memset(&cam, 0, sizeof cam);
  const double T[12] = {1,0,0, 0,1,0, 0,0,1, 0,0,0};
  std::vector<mcp_td_in> td;
  auto out = a.SearchForPoints(cam, T, T, td, 10, 8);
  CHECK(out.empty());
  std::vector<mcp_pose_point> pts; double bfw[12]; std::memcpy(bfw, T, sizeof T);
  auto mu2 = mcptam_hip::TrackMapPoseIterations(pts, { cam }, std::vector<double>(T, T + 12), bfw, { 1 }, { -1.0 });
  (void)mu2; (void)&KeyFrame::SE3fromSE2;
  // the cameras of a frame in one submission: identical to the per-camera calls; and what the submission costs a native caller
  {
    KeyFrame c1(w, h), c2(w, h), c3(w, h), c4(w, h);
    std::vector<KeyFrame*> kfs = { &c1, &c2, &c3, &c4 };
    KeyFrame::MakeKeyFrameLiteBatch(kfs, { img.data(), img.data(), img.data(), img.data() }, { w, w, w, w });
    for (KeyFrame* k : kfs) for (int l = 0; l < MCP_LEVELS; ++l) {
      mcptam_hip::Level X = k->GetLevel(l), Y = b.GetLevel(l);
      CHECK(X.image == Y.image && X.vCorners.size() == Y.vCorners.size() && X.nFastThresh == Y.nFastThresh && X.vCornerRowLUT == Y.vCornerRowLUT);
      for (size_t i = 0; i < X.vCorners.size(); ++i) CHECK(X.vCorners[i].x == Y.vCorners[i].x && X.vCorners[i].y == Y.vCorners[i].y);
    }
    std::vector<std::vector<mcp_td_in>> none(4);
    auto outs = KeyFrame::SearchForPointsBatch(kfs, { cam, cam, cam, cam }, T, std::vector<double>(48, 0.0), none, 10, 8);
    CHECK(outs.size() == 4 && outs[0].empty());
    const int reps = 50;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) KeyFrame::MakeKeyFrameLiteBatch(kfs, { img.data(), img.data(), img.data(), img.data() }, { w, w, w, w });
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()/reps;
    std::printf("native MakeKeyFrame_Lite of 4 cameras %dx%d in one submission (upload included): %.0f us\n", w, h, us);
  }
  std::printf("keyframe mirror ok: %zu corners, %zu patches matched\n", total_corners, pm.size());
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::strcmp(argv[1], "--link-only") == 0) {
    std::printf("linked: %d gfx950 device(s) visible\n", mcp_device_count());
    return (void*)&exercise ? 0 : 1;
  }
  try { const int rc = exercise(320, 240); return rc ? rc : exercise(640, 480); }
  catch (const std::exception& e) { std::printf("failed: %s\n", e.what()); return 2; }
}
