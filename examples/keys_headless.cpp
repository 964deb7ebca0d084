// The loop every run of the reference ends in (skeleton.cpp:226-230: `while( Update() ) { DrawStandard(...);
// SDL_Renderframe(screen); }`) written against raytracer_amd/host/photon_beams_host.h and driven by a script of key
// presses instead of a window: one key is queued before each Update, the frame after it is saved as <prefix>_<k>.bmp,
// and the script ends with SDLK_ESCAPE.  A key that moves the light is followed by a new photon pass (the reference
// casts once, before its loop; `cast` in the script asks for it).
//   ./keys_headless <photons> <width> <prefix> <key> [<key> ...]
//   keys: up down left right w s a d l j o u k i r cast
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../raytracer_amd/host/photon_beams_host.h"

using namespace pbhost;

static int key_of(const std::string& k) {
  if (k == "up") return SDLK_UP;
  if (k == "down") return SDLK_DOWN;
  if (k == "left") return SDLK_LEFT;
  if (k == "right") return SDLK_RIGHT;
  if (k.size() == 1 && std::strchr("wsadljoukir", k[0])) return (int)k[0];
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 5) { std::cerr << "usage: keys_headless <photons> <width> <prefix> <key> [<key> ...]" << std::endl; return 2; }
  try {
    UseDevicesFromEnvironment();
    control.photon_number = std::atoi(argv[1]);
    control.screen_width = control.screen_height = std::atoi(argv[2]);
    const std::string prefix = argv[3];
    std::vector<Triangle> triangles;
    LoadTestModel(triangles);
    std::vector<PhotonBeam> beams_const, beams_scatt;
    std::vector<PhotonSeg> items;
    auto cast = [&]() {
      mat4 matrix;
      TransformationMatrix(matrix);
      beams_const.clear(); beams_scatt.clear(); items.clear();
      AABB root_aabb = CastPhotonBeams(control.photon_number, beams_const, beams_scatt, matrix, triangles);
      BoundPhotonBeams(beams_const, items, triangles);
      BoundPhotonBeams(beams_scatt, items, triangles);
      Node* root = newNode(root_aabb);
      BuildTree(root, items, 0);
    };
    cast();
    screen* scr = InitializeSDL(control.screen_width, control.screen_height, false);
    int frame = 0, next = 4;
    for (;;) {
      if (next < argc) {
        const std::string k = argv[next++];
        if (k == "cast") cast();
        else if (int code = key_of(k)) PushKey(code);
        else { std::cerr << "unknown key " << k << std::endl; return 2; }
      } else PushKey(SDLK_ESCAPE);
      if (!Update()) break;
      DrawStandard(scr, beams_const, beams_scatt);
      SDL_Renderframe(scr);
      SDL_SaveImage(scr, (prefix + "_" + std::to_string(frame++) + ".bmp").c_str());
    }
    std::cout << "camera " << camera.x << " " << camera.y << " " << camera.z << " theta " << theta.x << " " << theta.y << " " << theta.z
              << " light " << light_position.x << " " << light_position.y << " " << light_position.z << std::endl;
    KillSDL(scr);
    Shutdown();
  } catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return 1; }
  return 0;
}
