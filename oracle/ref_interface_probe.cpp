// oracle/ref_interface_probe.cpp -- TEST INFRASTRUCTURE.  Compiles the ONE translation unit of the reference
// that builds without third-party libraries, frame_main/libs/MVS/Interface.h (the .mvs scene format, MVSI v5, and
// the raw 'DR' depth-map header), from where it lies under /root/reference, so that this repo's file-format code
// can be checked against the reference's own serializer.  Output binary: oracle/_ref/interface_probe (git-ignored).
//
//   interface_probe sizes                 -> prints sizeof(HeaderDepthDataRaw) and the 'DR' magic
//   interface_probe write out.mvs N M     -> writes a deterministic scene: N pinhole cameras/images, M vertices
//   interface_probe dump in.mvs           -> prints every field of the scene as text (%.17g)
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <vector>
#define _USE_CUSTOM_CV
#include "Interface.h"

using namespace _INTERFACE_NAMESPACE;

static int do_write(const char* path, int N, int M) {
	Interface scene;
	Interface::Platform platform;
	platform.name = "probe";
	for (int i = 0; i < N; ++i) {
		Interface::Platform::Camera cam;
		cam.name = "cam" + std::to_string(i);
		cam.width = 640 + 16 * i; cam.height = 480 + 8 * i;
		cam.K = Interface::Mat33d::eye();
		cam.K(0, 0) = 500.0 + i; cam.K(1, 1) = 501.0 + i; cam.K(0, 2) = 319.5; cam.K(1, 2) = 239.5;
		cam.R = Interface::Mat33d::eye();
		cam.C = Interface::Pos3d(0, 0, 0);
		platform.cameras.push_back(cam);
		Interface::Platform::Pose pose;
		pose.R = Interface::Mat33d::eye();
		pose.R(0, 0) = 0.0; pose.R(0, 1) = -1.0; pose.R(1, 0) = 1.0; pose.R(1, 1) = 0.0; // 90 degrees about z
		pose.C = Interface::Pos3d(0.25 * i, -0.125 * i, 0.0625 * i);
		platform.poses.push_back(pose);
		Interface::Image img;
		img.name = "img" + std::to_string(1000 + i) + ".png";
		img.platformID = 0; img.cameraID = (uint32_t)i; img.poseID = (uint32_t)i; img.ID = (uint32_t)i;
		scene.images.push_back(img);
	}
	scene.platforms.push_back(platform);
	for (int v = 0; v < M; ++v) {
		Interface::Vertex vert;
		vert.X = Interface::Pos3f(0.5f * v, 1.0f - 0.25f * v, 5.0f + v);
		for (int k = 0; k < 2 + (v % 2); ++k) {
			Interface::Vertex::View view;
			view.imageID = (uint32_t)((v + k) % N); view.confidence = 0.5f + 0.125f * k;
			vert.views.push_back(view);
		}
		scene.vertices.push_back(vert);
		scene.verticesColor.push_back(Interface::Color{Interface::Col3((uint8_t)(10 + v), (uint8_t)(20 + v), (uint8_t)(30 + v))});
	}
	return ARCHIVE::SerializeSave(scene, path) ? 0 : 2;
}

static int do_dump(const char* path) {
	Interface scene;
	uint32_t ver = 0;
	if (!ARCHIVE::SerializeLoad(scene, path, &ver)) return 2;
	printf("version %u\nplatforms %zu\n", ver, scene.platforms.size());
	for (const auto& p : scene.platforms) {
		printf("platform %s cameras %zu poses %zu\n", p.name.c_str(), p.cameras.size(), p.poses.size());
		for (const auto& c : p.cameras) {
			printf("camera %s %u %u K", c.name.c_str(), c.width, c.height);
			for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) printf(" %.17g", c.K(i, j));
			printf(" R");
			for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) printf(" %.17g", c.R(i, j));
			printf(" C %.17g %.17g %.17g\n", c.C.x, c.C.y, c.C.z);
		}
		for (const auto& q : p.poses) {
			printf("pose R");
			for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) printf(" %.17g", q.R(i, j));
			printf(" C %.17g %.17g %.17g\n", q.C.x, q.C.y, q.C.z);
		}
	}
	printf("images %zu\n", scene.images.size());
	for (const auto& im : scene.images) printf("image %s %u %u %u %u\n", im.name.c_str(), im.platformID, im.cameraID, im.poseID, im.ID);
	printf("vertices %zu\n", scene.vertices.size());
	for (const auto& v : scene.vertices) {
		printf("vertex %.9g %.9g %.9g views", v.X.x, v.X.y, v.X.z);
		for (const auto& w : v.views) printf(" %u:%.9g", w.imageID, w.confidence);
		printf("\n");
	}
	printf("colors %zu\n", scene.verticesColor.size());
	for (const auto& c : scene.verticesColor) printf("color %u %u %u\n", (unsigned)c.c.x, (unsigned)c.c.y, (unsigned)c.c.z);
	printf("normals %zu\n", scene.verticesNormal.size());
	return 0;
}

int main(int argc, char** argv) {
	if (argc >= 2 && !strcmp(argv[1], "sizes")) {
		printf("sizeof(HeaderDepthDataRaw) %zu magic 0x%04x\n", sizeof(HeaderDepthDataRaw), (unsigned)HeaderDepthDataRaw::HeaderDepthDataRawName());
		return 0;
	}
	if (argc >= 5 && !strcmp(argv[1], "write")) return do_write(argv[2], atoi(argv[3]), atoi(argv[4]));
	if (argc >= 3 && !strcmp(argv[1], "dump")) return do_dump(argv[2]);
	fprintf(stderr, "usage: interface_probe sizes | write out.mvs N M | dump in.mvs\n");
	return 1;
}
