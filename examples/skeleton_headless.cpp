// StandardRun (skeleton.cpp:205-233) and HeterogeneousRun (skeleton.cpp:111-203) of the reference written
// against raytracer_amd/host/photon_beams_host.h: the same calls in the same order, one frame,
// screenshot.bmp written headless.
//   ./skeleton_headless [photons] [bounces] [width] [out.bmp] [het]
//   ./skeleton_headless analysis      RUN_ANALYSIS (skeleton.cpp:82-84): NSD.csv and rRMSE.csv, no frame
//   ./skeleton_headless segments <file> [out.bmp]   DrawStandard over a dumped segment set instead of a photon pass
//                                      (tests/golden/cfg1_tree_segments.bin gives the reference's own screenshot)
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "../raytracer_amd/host/photon_beams_host.h"

using namespace pbhost;

int main(int argc, char** argv) {
  try { UseDevicesFromEnvironment(); } catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return 1; }
  if (argc > 1 && std::strcmp(argv[1], "analysis") == 0) {
    try { RunAnalysis(); Shutdown(); } catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return 1; }
    return 0;
  }
  if (argc > 2 && std::strcmp(argv[1], "segments") == 0) {
    try {
      std::vector<Triangle> triangles;
      LoadTestModel(triangles);
      mat4 matrix;
      TransformationMatrix(matrix);
      std::vector<PhotonBeam> none;
      LoadSegments(argv[2], matrix, triangles);
      screen* scr = InitializeSDL(control.screen_width, control.screen_height, false);
      DrawStandard(scr, none, none);
      SDL_Renderframe(scr);
      SDL_SaveImage(scr, argc > 3 ? argv[3] : "screenshot.bmp");
      KillSDL(scr);
      Shutdown();
    } catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return 1; }
    return 0;
  }
  if (argc > 1) control.photon_number = std::atoi(argv[1]);
  if (argc > 2) control.bounces = std::atoi(argv[2]);
  if (argc > 3) control.screen_width = control.screen_height = std::atoi(argv[3]);
  const char* out = argc > 4 ? argv[4] : "screenshot.bmp";
  const bool het = argc > 5 && std::atoi(argv[5]) != 0;
  control.heterogeneous = het ? 1 : 0;      // Control.h: HETEROGENEOUS, ONE_DIMENSIONAL
  control.one_dimensional = het ? 1 : 0;
  try {
    std::vector<Triangle> triangles;
    LoadTestModel(triangles);
    mat4 matrix;
    TransformationMatrix(matrix);
    if (het) {
      std::cout << "Creating grid" << std::endl;
      CreateSurface(27, -3.0, 0.016);
      float minimum = 10, maximum = -10;
      for (size_t i = 0; i < GRID.geometric_points.size(); i++) {
        pb_transform_points(&matrix.m[0][0], &GRID.geometric_points[i].x, 1);
        if (GRID.geometric_points[i].z < minimum) minimum = GRID.geometric_points[i].z;
        if (GRID.geometric_points[i].z > maximum) maximum = GRID.geometric_points[i].z;
      }
      std::cout << "Minimum value: " << minimum << std::endl;
      std::cout << "Maximum value: " << maximum << std::endl;
    }
    std::vector<PhotonBeam> beams_const, beams_scatt;
    std::vector<PhotonSeg> items;
    std::cout << "Casting photons" << std::endl;
    AABB root_aabb = CastPhotonBeams(control.photon_number, beams_const, beams_scatt, matrix, triangles);
    BoundPhotonBeams(beams_const, items, triangles);
    BoundPhotonBeams(beams_scatt, items, triangles);
    std::cout << "Beams size: " << (beams_const.size() + beams_scatt.size()) << "\n";
    std::cout << "Segment size: " << items.size() << "\n";
    Node* root = newNode(root_aabb);
    std::cout << "Building tree" << std::endl;
    BuildTree(root, items, 0);
    std::cout << "Calculating radiance" << std::endl;
    screen* scr = InitializeSDL(control.screen_width, control.screen_height, false);
    auto t0 = std::chrono::steady_clock::now();
    if (het) DrawHet(scr, beams_const, beams_scatt, nullptr);
    else DrawStandard(scr, beams_const, beams_scatt);
    SDL_Renderframe(scr);
    auto t1 = std::chrono::steady_clock::now();
    std::cout << "Render time: " << std::chrono::duration<double, std::milli>(t1 - t0).count() << " ms." << std::endl;
    pb_stats st = Stats();
    std::cout << "GPU frame " << st.ms_render << " ms, photon pass " << st.ms_emit << " ms, tree " << st.ms_build << " ms\n";
    SDL_SaveImage(scr, out);
    KillSDL(scr);
    delete root;
    Shutdown();
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
