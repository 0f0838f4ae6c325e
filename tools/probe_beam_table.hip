// Debugging aid: prints entries of the beam model's hit-distance table.
#include "../beluga_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
int main() {
  const mcl::BeamModel m{0.5, 0.5, 0.05, 0.05, 0.2, 0.1, 40.0};
  const uint32_t entries = mcl::beam_table_entries(40.0, 0.05);
  double* d;
  hipMalloc(&d, entries * 32ull);
  mcl::launch_beam_table(nullptr, m, 0.05, entries, d);
  std::vector<double> h(4ull * entries);
  hipMemcpy(h.data(), d, entries * 32ull, hipMemcpyDeviceToHost);
  for (uint32_t r2 : {0u, 1u, 2u, 4u, 5u, 100u, 640000u, entries - 2, entries - 1})
    std::printf("r2 %u: z %.17g hit %.17g short %.17g\n", r2, h[4ull * r2], h[4ull * r2 + 1], h[4ull * r2 + 2]);
  std::printf("entries %u\n", entries);
  return 0;
}
