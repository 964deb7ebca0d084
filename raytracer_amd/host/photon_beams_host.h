// C++ host layer with the reference's own function names, argument lists and globals for the photon-beam
// path, implemented on the C ABI of include/photon_beams.h.  A program written against the reference's
// headers (skeleton.cpp's StandardRun) compiles against this header instead of CastPhotons.h /
// BeamRadiance.h / SDLauxiliary.h and runs the path on the GPU.  Vector types are plain structs with
// glm's memory layout (column-major matrices); no glm, no SDL.
//
// Differences a caller can observe, all stated where they arise:
//  * CastPhotonBeams draws counter-based random numbers per photon instead of one std::mt19937 stream
//    (CastPhotons.h:86): same distribution, different beams for a given SEED.
//  * BuildTree builds a linear BVH; Node is an opaque handle (the reference's tree only influences the
//    image through each segment's gate box, which is PhotonSeg.min/max here).
//  * The display functions are headless: InitializeSDL allocates the ARGB buffer, SDL_Renderframe does
//    nothing, SDL_SaveImage writes a 32-bit BMP.
//  * Errors: the reference prints and exit(1)s only in its SDL code; here every failing GPU call throws
//    std::runtime_error carrying pb_last_error().
#ifndef PHOTON_BEAMS_HOST_H
#define PHOTON_BEAMS_HOST_H
#include <cstdint>
#include <vector>

#include "../../include/photon_beams.h"

namespace pbhost {

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct dvec3 { double x, y, z; };
struct mat4 { float m[4][4]; };   // m[col][row], as glm::mat4

// Control.h:40-50, skeleton.cpp:25 — the mutable globals the reference's functions read.
extern vec4 light_position;
extern vec3 light_power;
extern float absorption_c, scattering_c, extinction_c;
extern vec4 camera;
extern vec3 theta;
// Control.h:4-31 — compile-time switches there, run-time values here (same names, lower case).
extern pb_control control;
// GPUs a frame is spread over (HIP device indices; default {0}; skeleton_headless reads PB_DEVICES="0,1,...").  The
// photon pass runs on the first; every device holds the scene, the grid and the rendered segment set, and
// DrawStandard / DrawHet / Draw give the frame's rows to the devices in bands of 64, round robin, one host thread
// per device.  The image does not depend on the list (a device may be named twice).
extern std::vector<int> devices;
void UseDevicesFromEnvironment();   // devices = the list in PB_DEVICES, if that is set

struct Triangle {                 // TestModelH.h:12-39
  vec4 v0, v1, v2, normal;
  vec3 colour;
  bool water, reflective;
};
struct Intersection { vec4 position; float distance; int index; bool water; };   // Intersections.h:7-13
struct AABB { vec4 max, min, mid; };                                             // Intersections.h:28-33 (z flipped)
struct PhotonSeg {                                                               // Intersections.h:36-52
  vec4 start, end, mid, min, max, orig_start;
  float radius; bool ada_width; int id; float s_ext, e_ext, c_ext; int seg_id; bool scattered;
};
struct PhotonBeam {                                                              // CastPhotons.h:14-27
  vec4 start, end, omega_u, omega_v;
  bool ada_width; float offset, radius; dvec3 energy; int index_hit; bool absorbed, scattered;
};
struct Node { AABB aabb; bool leaf; };   // CastPhotons.h:49-56: handle only, the hierarchy lives on the GPU
struct screen { int height, width; uint32_t* buffer; };                          // SDLauxiliary.h:9-16
struct vec2 { float x, y; };
struct Grid { float height; int side_points; std::vector<vec4> geometric_points; };   // Tessendorf.h:16-21
struct NodeGen;                                                                  // CastPhotons.h:40-47: never dereferenced
extern Grid GRID;                                                                // Tessendorf.h:27

void LoadTestModel(std::vector<Triangle>& triangles);                            // TestModelH.h:45
void TransformationMatrix(mat4& M);                                              // skeleton.cpp:512

AABB CastPhotonBeams(int number, std::vector<PhotonBeam>& beams_const, std::vector<PhotonBeam>& beams_scattered,
                     const mat4& matrix, const std::vector<Triangle>& triangles);   // CastPhotons.h:99-103
// CastPhotons.h:290-310 as ProduceStopMotion uses it (skeleton.cpp:321-322): casts the beams of the first list of
// the last CastPhotonBeams again through the current GRID and rebuilds `beams_scattered`; `beams_const` (the
// caller's copy) is left as it is, as the reference discards what it appends to it.  The reference's loop
// runs over the vector it appends to and does not terminate; exactly the beams present on entry are re-cast.
AABB RecastPhotonBeams(std::vector<PhotonBeam>& beams_const, std::vector<PhotonBeam>& beams_scattered,
                       const mat4& matrix, const std::vector<Triangle>& triangles);
// Appends the segments of `beams` (the list of that size returned by the last CastPhotonBeams).
void BoundPhotonBeams(std::vector<PhotonBeam>& beams, std::vector<PhotonSeg>& items,
                      const std::vector<Triangle>& triangles);                       // CastPhotons.h:107
Node* newNode(AABB data);                                                            // CastPhotons.h:58
// Makes `child` the segment set to render (gate box = PhotonSeg.min/max, energy looked up as
// BeamRadiance.h:449-455 does) and builds the acceleration structure over it.
void BuildTree(Node* parent, std::vector<PhotonSeg>& child, int leftover);           // CastPhotons.h:105

// Tessendorf.h:47-75: fills GRID (host arithmetic, pb_create_surface).  HeterogeneousRun then multiplies
// every GRID point by the camera matrix itself (skeleton.cpp:119-125); CastPhotonBeams hands GRID's z
// values to the GPU when control.heterogeneous is set.
void CreateSurface(int triangle_number, float height, double time);
// Heterogeneous.h:29-135 for one ray (GPU; use pb_extinction3d directly for batches).
float Extinction3D(const vec4 start, const vec4 end, const Grid grid, std::vector<vec2>& dist_ext);

screen* InitializeSDL(int width, int height, bool fullscreen = false);               // SDLauxiliary.h:72
void PutPixelSDL(screen* s, int x, int y, vec3 colour);                              // SDLauxiliary.h:149
void SDL_Renderframe(screen* s);                                                     // SDLauxiliary.h:64
void SDL_SaveImage(screen* s, const char* filename);                                 // SDLauxiliary.h:24
void KillSDL(screen* s);                                                             // SDLauxiliary.h:55

// One frame of skeleton.cpp:330-365 on the GPU; uses the tree of the last BuildTree as the reference
// uses its global `root`.
void DrawStandard(screen* screen, std::vector<PhotonBeam> beams_const, std::vector<PhotonBeam> beams_scatt);

// skeleton.cpp:414-473: BeamRadiance's per-segment dispatch (BeamRadiance.h:486-543) — cones for adaptive-width
// segments, else the estimator control.one_dimensional and control.heterogeneous select; control.short_view.
void Draw(screen* screen, std::vector<PhotonBeam> beams_const, std::vector<PhotonBeam> beams_scattered,
          std::vector<PhotonSeg>& items);
// One frame of skeleton.cpp:367-412 (BVHBeamRadianceHet with Integral_73Het); the NodeGen argument is
// unused there as well.
void DrawHet(screen* screen, std::vector<PhotonBeam> beams_const, std::vector<PhotonBeam> beams_scattered, NodeGen*);

// Analysis.h:81-122: writes NSD.csv and rRMSE.csv into the current directory, in the reference's format
// (analysis_extinction = EXT_ANALYSIS = 0.04f widened, Control.h:21, Analysis.h:15).
void RunAnalysis();

// Not in the reference: the segment set as a file (pb_save_segments / pb_load_segments).  LoadSegments stands in for
// CastPhotonBeams + BoundPhotonBeams + BuildTree when a set dumped elsewhere is to be drawn (it takes the matrix and
// the triangles those calls would have been given); SaveSegments dumps what the next Draw* would use.
void LoadSegments(const char* path, const mat4& matrix, const std::vector<Triangle>& triangles);
void SaveSegments(const char* path);

// skeleton.cpp:475-486 and :534-610, the loop every run of the reference ends in (`while( Update() ) { Draw...; }`),
// without SDL: Update prints the time since its last call ("Render time: ... ms.") and returns UserInput(); UserInput
// applies every queued key press to theta / camera / light_position exactly as the reference's switch does and returns
// false on SDLK_ESCAPE or a queued quit.  Keys come from PushKey (the SDLK_* values below are SDL2's) instead of a
// window; an empty queue leaves the state alone and returns true.  As in the reference, Draw* read
// TransformationMatrix from the globals on every frame (so a camera key moves the walls of the next frame), while the
// beams stay as cast until the caller casts again (a light key only matters to the next CastPhotonBeams).
enum : int {
  SDLK_ESCAPE = 27, SDLK_a = 'a', SDLK_d = 'd', SDLK_i = 'i', SDLK_j = 'j', SDLK_k = 'k', SDLK_l = 'l', SDLK_o = 'o',
  SDLK_r = 'r', SDLK_s = 's', SDLK_u = 'u', SDLK_w = 'w',
  SDLK_RIGHT = 0x4000004F, SDLK_LEFT = 0x40000050, SDLK_DOWN = 0x40000051, SDLK_UP = 0x40000052
};
void PushKey(int sdl_key_code);     // an SDL_KEYDOWN event for the next UserInput
void PushQuit();                    // an SDL_QUIT event
bool UserInput();                   // skeleton.cpp:534-610
bool Update();                      // skeleton.cpp:475-486

// Counters and device times of the last calls (pb_counters).
pb_stats Stats();
// Releases the GPU context (the reference leaks its tree; a long-running host may not want to).
void Shutdown();

}  // namespace pbhost
#endif
