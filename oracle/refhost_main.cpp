// refhost_main.cpp -- TEST INFRASTRUCTURE (oracle/): the reference host's main() / runCuda() / saveImage()
// sequence (src/main.cpp:33-76, 78-99, 101-147) without GLFW / GL, compiled together with the REFERENCE'S OWN
// scene.cpp, utilities.cpp, image.cpp and stb.cpp (where they lie under /root/reference) and this repository's
// reference-side binding host/pathtrace_shim.cpp, and linked against libptmi355.so:
//
//     oracle/_ref/refhost SCENEFILE.txt TIMESTRING
//
// It is what a maintainer gets after the INTEGRATION.md edit, minus the preview window: the reference's loader
// fills Scene, runCuda's camera recompute runs on the first frame, pathtraceFree() is called BEFORE the first
// pathtraceInit() (main.cpp:126), pathtrace(pbo = NULL, 0, iteration) refreshes scene->state.image on every call,
// and saveImage() writes <FILE>.<TIMESTRING>.<N>samp.png through the reference's image class.
// tests/test_gpu_parity.py::test_reference_host_through_the_shim compares that PNG with ptbench's.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

#include <glm/glm.hpp>
#include <glm/gtx/transform.hpp>

#include "image.h"
#include "pathtrace.h"
#include "scene.h"
#include "sceneStructs.h"
#include "utilities.h"

using namespace std;

static std::string startTimeString;
static bool camchanged = true;                       // main.cpp:14
float zoom, theta, phi;                              // main.cpp:18-20
glm::vec3 cameraPosition;
glm::vec3 ogLookAt;
Scene *scene;
RenderState *renderState;
int iteration;
int width;
int height;

static void saveImage() {                            // main.cpp:78-99
    float samples = iteration;
    image img(width, height);
    for (int x = 0; x < width; x++) {
        for (int y = 0; y < height; y++) {
            int index = x + (y * width);
            glm::vec3 pix = renderState->image[index];
            img.setPixel(width - 1 - x, y, glm::vec3(pix) / samples);
        }
    }
    std::string filename = renderState->imageName;
    std::ostringstream ss;
    ss << filename << "." << startTimeString << "." << samples << "samp";
    filename = ss.str();
    img.savePNG(filename);
}

static bool runCuda() {                              // main.cpp:101-147; false = the render is complete
    if (camchanged) {
        iteration = 0;
        Camera &cam = renderState->camera;
        cameraPosition.x = zoom * sin(phi) * sin(theta);
        cameraPosition.y = zoom * cos(theta);
        cameraPosition.z = zoom * cos(phi) * sin(theta);
        cam.view = -glm::normalize(cameraPosition);
        glm::vec3 v = cam.view;
        glm::vec3 u = glm::vec3(0, 1, 0);
        glm::vec3 r = glm::cross(v, u);
        cam.up = glm::cross(r, v);
        cam.right = r;
        cam.position = cameraPosition;
        cameraPosition += cam.lookAt;
        cam.position = cameraPosition;
        camchanged = false;
    }
    if (iteration == 0) {
        pathtraceFree();
        pathtraceInit(scene);
    }
    if (iteration < renderState->iterations) {
        uchar4 *pbo_dptr = NULL;                     // no GL on a headless MI355X node: the shim takes NULL
        iteration++;
        int frame = 0;
        pathtrace(pbo_dptr, frame, iteration);
        return true;
    }
    saveImage();
    pathtraceFree();
    return false;
}

int main(int argc, char **argv) {                    // main.cpp:33-76
    if (argc < 3) {
        printf("Usage: %s SCENEFILE.txt TIMESTRING\n", argv[0]);
        return 1;
    }
    startTimeString = argv[2];
    scene = new Scene(argv[1]);
    iteration = 0;
    renderState = &scene->state;
    Camera &cam = renderState->camera;
    width = cam.resolution.x;
    height = cam.resolution.y;
    glm::vec3 view = cam.view;
    glm::vec3 up = cam.up;
    glm::vec3 right = glm::cross(view, up);
    up = glm::cross(right, view);
    cameraPosition = cam.position;
    glm::vec3 viewXZ = glm::vec3(view.x, 0.0f, view.z);
    glm::vec3 viewZY = glm::vec3(0.0f, view.y, view.z);
    phi = glm::acos(glm::dot(glm::normalize(viewXZ), glm::vec3(0, 0, -1)));
    theta = glm::acos(glm::dot(glm::normalize(viewZY), glm::vec3(0, 1, 0)));
    ogLookAt = cam.lookAt;
    zoom = glm::length(cam.position - ogLookAt);
    (void)up;
    while (runCuda()) {}                             // mainLoop(): runCuda() once per frame
    return 0;
}
