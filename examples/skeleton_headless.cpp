// StandardRun of the reference (skeleton.cpp:205-233) written against raytracer_amd/host/photon_beams_host.h:
// the same calls in the same order, one frame, screenshot.bmp written headless.
//   ./skeleton_headless [photons] [bounces] [width] [out.bmp]
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "../raytracer_amd/host/photon_beams_host.h"

using namespace pbhost;

int main(int argc, char** argv) {
  if (argc > 1) control.photon_number = std::atoi(argv[1]);
  if (argc > 2) control.bounces = std::atoi(argv[2]);
  if (argc > 3) control.screen_width = control.screen_height = std::atoi(argv[3]);
  const char* out = argc > 4 ? argv[4] : "screenshot.bmp";
  try {
    std::vector<Triangle> triangles;
    LoadTestModel(triangles);
    mat4 matrix;
    TransformationMatrix(matrix);
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
    DrawStandard(scr, beams_const, beams_scatt);
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
