// api_demo.cpp — the reference's documented rasterizer usage (README.md:127-205 of Liupeter01/Software-Rasterizer),
// written against OUR headers.  Exit codes: 0 ok, 3 = no GPU (constructor threw), 1 = wrong output.
#include <SoftRasterizer.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

int main(int argc, char **argv) {
  const std::string home = (argc > 1 ? std::string(argv[1]) : std::string(".")) + "/assets/";
  try {
    auto render = std::make_shared<SoftRasterizer::TraditionalRasterizer>(256, 256);
    auto scene = std::make_shared<SoftRasterizer::Scene>("TestScene", glm::vec3(0.0f, 0.0f, 0.9f), glm::vec3(0.0f, 0.0f, 0.0f),
                                                         glm::vec3(0.0f, 1.0f, 0.0f));
    float degree = 30.0f;
    if (!scene->addGraphicObj(home + "models/spot/spot_triangulated_good.obj", "spot", glm::vec3(0, 1, 0), degree,
                              glm::vec3(0.f, 0.0f, 0.0f), glm::vec3(0.3f, 0.3f, 0.3f)))
      return 1;
    if (!scene->addShader("spot_shader", home + "models/spot/spot_texture.png", SoftRasterizer::SHADERS_TYPE::TEXTURE)) return 1;
    if (!scene->startLoadingMesh("spot")) return 1;
    if (!scene->bindShader2Mesh("spot", "spot_shader")) return 1;
    auto light1 = std::make_shared<SoftRasterizer::light_struct>();
    light1->position = glm::vec3{0.9, 0.9, -0.9f};
    light1->intensity = glm::vec3{100, 100, 100};
    auto light2 = std::make_shared<SoftRasterizer::light_struct>();
    light2->position = glm::vec3{0.f, 0.8f, 0.9f};
    light2->intensity = glm::vec3{50, 50, 50};
    scene->addLight("Light1", light1);
    scene->addLight("Light2", light2);
    if (!render->addScene(scene)) return 1;
    if (render->addScene(scene)) return 1; // duplicate → false

    render->clear(SoftRasterizer::Buffers::Color | SoftRasterizer::Buffers::Depth);
    scene->setModelMatrix("spot", glm::vec3(0.f, 1.f, 0.f), degree, glm::vec3(0.0f), glm::vec3(0.3f));
    scene->setViewMatrix(glm::vec3(0.0f, 0.0f, 0.9f), glm::vec3(0.0f, 0.0f, 0.0f), glm::vec3(0.0f, 1.0f, 0.0f));
    scene->setProjectionMatrix(45.0f, 0.1f, 100.0f);
    render->collect_stats = true;
    render->display(SoftRasterizer::Primitive::TRIANGLES);

    size_t covered = 0;
    for (float z : render->zBuffer()) covered += std::isfinite(z) ? 1 : 0;
    std::printf("covered=%zu fragments=%llu visible=%llu\n", covered, (unsigned long long)render->last_stats.fragments,
                (unsigned long long)render->last_stats.visible);
    if (covered == 0 || covered != render->last_stats.visible) return 1;
    // the resolved 8-bit image (device resolve) must equal the host-side rounding of the float planes
    const auto &bgr = render->frameBuffer8();
    for (size_t i = 0; i < 256 * 256; ++i)
      for (int c = 0; c < 3; ++c) {
        float v = render->channel(c)[i];
        long r = std::lrintf(v);
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        if (bgr[i * 3 + c] != (unsigned char)r) return 1;
      }
    // host vertex stage (loadTriangleStream on the CPU) must give the same planes, bit for bit
    auto render2 = std::make_shared<SoftRasterizer::TraditionalRasterizer>(256, 256);
    render2->device_vertex_stage = false;
    if (!render2->addScene(scene)) return 1;
    scene->setProjectionMatrix(45.0f, 0.1f, 100.0f);
    render2->clear(SoftRasterizer::Buffers::Color | SoftRasterizer::Buffers::Depth);
    render2->draw(SoftRasterizer::Primitive::TRIANGLES);
    if (std::memcmp(render2->zBuffer().data(), render->zBuffer().data(), 256 * 256 * 4) != 0) return 1;
    for (int c = 0; c < 3; ++c)
      if (std::memcmp(render2->channel(c).data(), render->channel(c).data(), 256 * 256 * 4) != 0) return 1;
    // draw() never clears: a second draw over the same buffers is idempotent; a depth-only clear keeps the colours
    render->draw(SoftRasterizer::Primitive::TRIANGLES);
    if (std::memcmp(render2->zBuffer().data(), render->zBuffer().data(), 256 * 256 * 4) != 0) return 1;
    render->clear(SoftRasterizer::Buffers::Depth);
    size_t inf = 0;
    for (float z : render->zBuffer()) inf += std::isinf(z) ? 1 : 0;
    if (inf != 256 * 256) return 1;
    if (std::memcmp(render2->channel(1).data(), render->channel(1).data(), 256 * 256 * 4) != 0) return 1;
    // ten frames of the reference's main loop (src/main.cpp:113-175): clear, set matrices, display
    for (int k = 0; k < 10; ++k) {
      degree += 10.0f;
      render->clear(SoftRasterizer::Buffers::Color | SoftRasterizer::Buffers::Depth);
      scene->setModelMatrix("spot", glm::vec3(0.f, 1.f, 0.f), degree, glm::vec3(0.0f), glm::vec3(0.3f));
      scene->setViewMatrix(glm::vec3(0.0f, 0.0f, 0.9f), glm::vec3(0.0f, 0.0f, 0.0f), glm::vec3(0.0f, 1.0f, 0.0f));
      scene->setProjectionMatrix(45.0f, 0.1f, 100.0f);
      render->display(SoftRasterizer::Primitive::TRIANGLES);
    }
    if (argc > 2) { // dump the last frame (degree = 130) as raw float planes z,c0,c1,c2 for the oracle comparison
      std::FILE *fp = std::fopen(argv[2], "wb");
      if (!fp) return 1;
      std::fwrite(render->zBuffer().data(), 4, 256 * 256, fp);
      for (int c = 0; c < 3; ++c) std::fwrite(render->channel(c).data(), 4, 256 * 256, fp);
      std::fclose(fp);
    }
    { // the reference re-reads the mesh on every draw() (src/Scene.cpp:927-947): an in-place edit of the vertices (same face
      // count, same object) must reach the GPU copy the device vertex stage keeps
      auto obj = scene->getMeshObj("spot");
      auto *mesh = obj ? dynamic_cast<SoftRasterizer::Mesh *>(obj->get()) : nullptr;
      if (!mesh) return 1;
      for (auto &v : mesh->vertices) v.position = v.position * 0.5f;
      for (auto r : {render, render2}) {
        r->clear(SoftRasterizer::Buffers::Color | SoftRasterizer::Buffers::Depth);
        scene->setProjectionMatrix(45.0f, 0.1f, 100.0f);
        r->draw(SoftRasterizer::Primitive::TRIANGLES);
      }
      size_t cov2 = 0;
      for (float z : render->zBuffer()) cov2 += std::isfinite(z) ? 1 : 0;
      std::printf("after the in-place edit: covered=%zu\n", cov2);
      if (cov2 == 0 || cov2 * 3 > covered * 2) return 1; // half the size: about a quarter of the pixels
      if (std::memcmp(render2->zBuffer().data(), render->zBuffer().data(), 256 * 256 * 4) != 0) return 1;
      for (int c = 0; c < 3; ++c)
        if (std::memcmp(render2->channel(c).data(), render->channel(c).data(), 256 * 256 * 4) != 0) return 1;
    }
    bool threw = false;
    try {
      render->draw(static_cast<SoftRasterizer::Primitive>(7));
    } catch (const std::runtime_error &e) {
      threw = std::string(e.what()) == "Primitive Type is not supported!";
    }
    return threw ? 0 : 1;
  } catch (const std::runtime_error &e) {
    std::fprintf(stderr, "runtime_error: %s\n", e.what());
    return 3;
  }
}
