// Scanner buffer API subset (scanner/util/memory.h): all element and frame memory is allocated
// through these so that the engine owns it after insert_element()/insert_frame().
// Reference uses: new_block_buffer_size (histogram_kernel_cpu.cpp:23), new_block_buffer
// (histogram_kernel_gpu.cpp:38-39), new_buffer/delete_buffer (montage_kernel_cpu.cpp:30-40),
// memcpy_buffer (info_from_frame_kernel.cpp:24-26), add_buffer_ref (pass_kernel.cpp:15-18).
#pragma once
#include "scanner/util/common.h"

namespace scanner {

u8* new_buffer(DeviceHandle device, size_t size);
// One allocation shared by `refs` elements; freed when the last reference is released.
u8* new_block_buffer(DeviceHandle device, size_t size, i32 refs);
inline u8* new_block_buffer_size(DeviceHandle device, size_t element_size, i32 count) {
  return new_block_buffer(device, element_size * (size_t)count, count);
}
void add_buffer_ref(DeviceHandle device, u8* buffer);
void add_buffer_refs(DeviceHandle device, u8* buffer, size_t refs);
void delete_buffer(DeviceHandle device, u8* buffer);  // releases one reference
void memcpy_buffer(u8* dest, DeviceHandle dest_device, const u8* src, DeviceHandle src_device, size_t size);
// shim-only introspection used by the tests: live allocations / references per device type
size_t shim_live_buffers(DeviceType type);
size_t shim_dev_pool_bytes();          // bytes idle in the device-buffer pool
size_t shim_dev_pool_drain(int device);  // give them back to the driver (all devices if < 0); bytes released

}  // namespace scanner
