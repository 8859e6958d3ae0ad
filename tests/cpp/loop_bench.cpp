// loop_bench.cpp — the reference README's own rasterization harness (README.md:619-642): 1024x1024, spot TEXTURE, 100 warm-up +
// 1000 timed frames, angle rotated each frame, clear() per frame, std::chrono around draw() (and, separately, display()).
// Written against OUR SoftRasterizer.hpp; prints one JSON line.  Single-frame LATENCY path (one frame in flight).
#include <SoftRasterizer.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

int main(int argc, char **argv) {
  const std::string home = (argc > 1 ? std::string(argv[1]) : std::string(".")) + "/assets/";
  const int frames = argc > 2 ? std::atoi(argv[2]) : 1000;
  auto render = std::make_shared<SoftRasterizer::TraditionalRasterizer>(1024, 1024);
  auto scene = std::make_shared<SoftRasterizer::Scene>("TestScene", glm::vec3(0.0f, 0.0f, 0.9f), glm::vec3(0.0f), glm::vec3(0.0f, 1.0f, 0.0f));
  scene->addGraphicObj(home + "models/spot/spot_triangulated_good.obj", "spot", glm::vec3(0, 1, 0), 0.f, glm::vec3(0.f), glm::vec3(0.3f));
  scene->addShader("spot_shader", home + "models/spot/spot_texture.png", SoftRasterizer::SHADERS_TYPE::TEXTURE);
  scene->startLoadingMesh("spot");
  scene->bindShader2Mesh("spot", "spot_shader");
  scene->addLight("Light1", std::make_shared<SoftRasterizer::light_struct>(glm::vec3{0.9, 0.9, -0.9f}, glm::vec3{100, 100, 100}));
  scene->addLight("Light2", std::make_shared<SoftRasterizer::light_struct>(glm::vec3{0.f, 0.8f, 0.9f}, glm::vec3{50, 50, 50}));
  render->addScene(scene);
  auto run = [&](bool with_display) {
    std::vector<double> ms;
    float degree = 0.f;
    for (int i = 0; i < 100 + frames; ++i) {
      render->clear(SoftRasterizer::Buffers::Color | SoftRasterizer::Buffers::Depth);
      scene->setModelMatrix("spot", glm::vec3(0.f, 1.f, 0.f), degree, glm::vec3(0.f), glm::vec3(0.3f));
      scene->setViewMatrix(glm::vec3(0.0f, 0.0f, 0.9f), glm::vec3(0.0f), glm::vec3(0.0f, 1.0f, 0.0f));
      scene->setProjectionMatrix(45.0f, 0.1f, 100.0f);
      auto t0 = std::chrono::high_resolution_clock::now();
      if (with_display)
        render->display(SoftRasterizer::Primitive::TRIANGLES); // draw + device resolve + 3 B/px read-back
      else {
        render->draw(SoftRasterizer::Primitive::TRIANGLES);
        render->frameBuffer8(); // (no read-back)
      }
      auto t1 = std::chrono::high_resolution_clock::now();
      if (i >= 100) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
      degree += 10.f;
      if (degree >= 360.f) degree = 0.f;
    }
    std::sort(ms.begin(), ms.end());
    return ms;
  };
  auto d = run(false);
  render->zBuffer(); // force completion of the last asynchronous draw
  auto p = run(true);
  auto q = [](const std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
  std::printf("{\"frames\": %d, \"draw_ms\": {\"median\": %.4f, \"p10\": %.4f, \"p90\": %.4f, \"min\": %.4f, \"max\": %.4f}, "
              "\"display_ms\": {\"median\": %.4f, \"p10\": %.4f, \"p90\": %.4f, \"min\": %.4f, \"max\": %.4f}, "
              "\"note\": \"draw() is asynchronous (device-resident framebuffer): its time is submission cost; display() includes device "
              "execution, the 8-bit resolve and the 3 MB read-back, i.e. a full frame of latency\"}\n",
              frames, q(d, .5), q(d, .1), q(d, .9), d.front(), d.back(), q(p, .5), q(p, .1), q(p, .9), p.front(), p.back());
  return 0;
}
