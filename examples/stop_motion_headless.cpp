// ProduceStopMotion of the reference (skeleton.cpp:265-328) written against
// raytracer_amd/host/photon_beams_host.h: beams cast once, then per frame the medium is replaced, the beams
// are cast again through it (RecastPhotonBeams), segmented, and a frame is drawn and saved.
//   ./stop_motion_headless [photons] [width] [begin] [finish] [prefix] [real_time]
// The reference passes `i/1000` (integer division, i.e. 0 for every frame) as the time of the new surface;
// real_time = 1 passes i/1000.0 instead so that the frames differ.
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <vector>

#include "../raytracer_amd/host/photon_beams_host.h"

using namespace pbhost;

static void surface(const mat4& matrix, double time) {
  CreateSurface(27, -3.0, time);
  for (size_t i = 0; i < GRID.geometric_points.size(); i++) pb_transform_points(&matrix.m[0][0], &GRID.geometric_points[i].x, 1);
}

int main(int argc, char** argv) {
  try { UseDevicesFromEnvironment(); } catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return 1; }
  if (argc > 1) control.photon_number = std::atoi(argv[1]);
  if (argc > 2) control.screen_width = control.screen_height = std::atoi(argv[2]);
  const int begin = argc > 3 ? std::atoi(argv[3]) : 16, finish = argc > 4 ? std::atoi(argv[4]) : 24;
  const std::string prefix = argc > 5 ? argv[5] : "screenshot_";
  const bool real_time = argc > 6 && std::atoi(argv[6]) != 0;
  control.heterogeneous = 1; control.one_dimensional = 1;
  try {
    std::vector<Triangle> triangles;
    LoadTestModel(triangles);
    std::vector<PhotonBeam> beams_const, beams_scattered;
    std::vector<PhotonSeg> items;
    mat4 matrix;
    TransformationMatrix(matrix);
    surface(matrix, 0.016);
    AABB root_aabb = CastPhotonBeams(control.photon_number, beams_const, beams_scattered, matrix, triangles);
    BoundPhotonBeams(beams_const, items, triangles);
    BoundPhotonBeams(beams_scattered, items, triangles);
    Node* root = newNode(root_aabb);
    BuildTree(root, items, 0);
    for (int i = begin; i < finish; i++) {
      screen* scr = InitializeSDL(control.screen_width, control.screen_height, false);
      Draw(scr, beams_const, beams_scattered, items);
      SDL_Renderframe(scr);
      std::ostringstream oss;
      oss << prefix << i << "_.bmp";
      SDL_SaveImage(scr, oss.str().c_str());
      KillSDL(scr);
      pb_stats st = Stats();
      std::cout << "frame " << i << ": " << (beams_const.size() + beams_scattered.size()) << " beams, " << items.size()
                << " segments, photon pass " << st.ms_emit << " ms, tree " << st.ms_build << " ms, frame " << st.ms_render
                << " ms" << std::endl;
      beams_scattered.clear();
      items.clear();
      surface(matrix, real_time ? i / 1000.0 : (double)(i / 1000));
      std::vector<PhotonBeam> beams_copy = beams_const;
      root_aabb = RecastPhotonBeams(beams_copy, beams_scattered, matrix, triangles);
      BoundPhotonBeams(beams_const, items, triangles);
      BoundPhotonBeams(beams_scattered, items, triangles);
      delete root;
      root = newNode(root_aabb);
      BuildTree(root, items, 0);
    }
    delete root;
    Shutdown();
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
