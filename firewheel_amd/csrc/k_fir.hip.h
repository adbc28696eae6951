// k_fir.hip.h — part of the single device translation unit fwgpu_kernels.hip (included inside namespace fwgpu).
// FIR convolution bank on the f32 matrix cores.
#pragma once

// ------------------------------------------------------------------ FIR convolution bank on the matrix cores
// SPEC ("fir"): y[n] = sum_k h[k] x[n-k].  Per block the outputs of all rows that share one
// impulse response are ONE dense GEMM:  Y[rows x frames] = Xwin[rows x W] * H[W x frames],  W = T-1+frames,
// Xwin[r][m] = x_r[n0-(T-1)+m] (history then the current block), H[m][i] = h[T-1-(m-i)] for 0 <= m-i <= T-1 else 0
// (Toeplitz, generated on the fly from h).  v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain, so the
// summation order is fully defined: the window is cut in segments of FIR_SEG positions, each segment is one
// fused chain in ascending m starting from +0.0, segment partials are added in segment order.  The oracle
// evaluates exactly that order with fmaf, so GPU == oracle bit for bit; vs an f64 convolution the error is the
// usual ~sqrt(W) * 2^-24 * sum|h x| (H7).
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_ir_convert(const SampleDesc* __restrict__ samples, int sample, int ch, float* __restrict__ dst, uint32_t T) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const SampleDesc sd = samples[sample];
    int c = ch < sd.channels ? ch : 0;  // a mono impulse response serves every channel
    dst[i] = i < sd.frames ? sample_fetch(sd, c, i) : 0.f;
}

// append the blocks' input to each row's mirrored history ring (positions q and q+R hold the same sample);
// blockIdx.y = block of the K-batch (the ring holds T-1 + K*max_block_frames samples: every block's window is there)
__global__ void k_fir_append(DevView v, const FirRow* __restrict__ rows, int n_rows) {
    int r = blockIdx.x;
    if (r >= n_rows) return;
    const FirRow row = rows[r];
    if (row.state < 0) return;  // padding row (tiles are impulse-response-homogeneous)
    const uint32_t kb = blockIdx.y;
    const NodeState* s = &v.states[row.state];
    const uint32_t R = (uint32_t)s->loop_end, p = (uint32_t)s->playhead;
    float* ring = v.ext + s->ext_off + (size_t)row.ch * 2u * R;
    const float* in = v.pool + (size_t)kb * v.pool_blk_stride + (size_t)row.in_buf * v.stride;
    for (int f = threadIdx.x; f < v.frames; f += blockDim.x) {
        uint32_t q = (p + kb * (uint32_t)v.frames + (uint32_t)f) % R;
        float x = in[f];
        ring[q] = x;
        ring[q + R] = x;
    }
}

#define FIR_PITCH (FIR_KC + 1)  // LDS row pitch in floats: 65 -> the 32 rows of a column hit 32 different banks
__global__ __launch_bounds__(256) void k_fir_gemm(DevView v, const FirRow* __restrict__ rows, int n_rows,
                                                  const uint32_t* __restrict__ tile_h_off, uint32_t T,
                                                  float* __restrict__ partials, int n_rows_pad, int n_pad, int col_groups) {
    __shared__ float lds[2 * 32 * FIR_PITCH + 2 * (256 + FIR_KC)];
    float* As = lds;                          // [2][32][FIR_PITCH]
    float* Hw = lds + 2 * 32 * FIR_PITCH;     // [2][256 + FIR_KC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 32;
    const uint32_t seg = blockIdx.y;
    const uint32_t kb = blockIdx.z / (uint32_t)col_groups;               // block of the K-batch
    const int ib = (int)(blockIdx.z % (uint32_t)col_groups) * 256;       // first output frame of this column group
    const int frames = v.frames;
    const uint32_t W = T - 1u + (uint32_t)frames;
    const uint32_t m_begin = seg * FIR_SEG;
    const uint32_t m_end = m_begin + FIR_SEG < W ? m_begin + FIR_SEG : W;
    const float* h = v.ext + tile_h_off[blockIdx.x];  // every row of a tile convolves with the same h

    // loader role: thread t stages 8 consecutive window positions of row (t >> 3)
    constexpr int NA = FIR_KC / 8;  // floats per loader thread: 8 threads cover one row of the chunk
    const int lrow = tid >> 3, lcol = (tid & 7) * NA;
    const float* wptr = nullptr;
    if (row0 + lrow < n_rows && rows[row0 + lrow].state >= 0) {
        const FirRow row = rows[row0 + lrow];
        const NodeState* s = &v.states[row.state];
        const uint32_t R = (uint32_t)s->loop_end, p = (uint32_t)s->playhead;
        const uint32_t e2 = (p + (kb + 1u) * (uint32_t)frames - 1u) % R + R;  // block kb's newest sample, upper mirror
        wptr = v.ext + s->ext_off + (size_t)row.ch * 2u * R + (e2 + 1u - W);
    }
    // Staging loads are unconditional and vectorised (addresses clamped into the ext pool, which carries 256 floats
    // of slack) and only ISSUED here; the selects that zero what lies outside the segment / the impulse response
    // touch the loaded registers — and therefore wait for them — in store_chunk, one MFMA loop later.  (A branch
    // per element would serialise eight HBM round trips per chunk; a select next to the load would expose one.)
    v4f xa[NA / 4];
#pragma unroll
    for (int j = 0; j < NA / 4; ++j) xa[j] = splat(0.f);
    float hraw0 = 0.f, hraw1 = 0.f;
    uint32_t m0_staged = 0;
    const float* wsafe = wptr ? wptr : v.ext;
    // Hw[q] = h[k], k = ib + T-1 - m0 - (KC-1) + q  (0 outside [0, T))
    auto h_index = [&](uint32_t m0, int q) -> long long {
        return (long long)ib + (long long)T - 1 - (long long)m0 - (FIR_KC - 1) + q;
    };
    auto h_clamp = [&](long long k) -> long long { return k < 0 ? 0 : (k >= (long long)T ? (long long)T - 1 : k); };
    auto load_chunk = [&](uint32_t m0) {
        m0_staged = m0;
#pragma unroll
        for (int j = 0; j < NA / 4; ++j) xa[j] = *(const v4f_u*)(wsafe + m0 + (uint32_t)(lcol + 4 * j));
        hraw0 = __builtin_nontemporal_load(h + h_clamp(h_index(m0, tid)));
        if (wave < FIR_KC / 64) hraw1 = __builtin_nontemporal_load(h + h_clamp(h_index(m0, tid + 256)));  // q = 256 .. 256+KC-1
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * 32 * FIR_PITCH + lrow * FIR_PITCH + lcol;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const uint32_t m = m0_staged + (uint32_t)(lcol + j);
            a[j] = (wptr && m < m_end) ? xa[j >> 2][j & 3] : 0.f;
        }
        float* hw = Hw + buf * (256 + FIR_KC);
        const long long k0 = h_index(m0_staged, tid), k1 = h_index(m0_staged, tid + 256);
        hw[tid] = (k0 >= 0 && k0 < (long long)T) ? hraw0 : 0.f;
        if (tid < FIR_KC) hw[tid + 256] = (k1 >= 0 && k1 < (long long)T) ? hraw1 : 0.f;
    };

    v16f acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = 0.f;
    const int ct0 = wave * 2, ct1 = wave * 2 + 1;  // this wave's two 32-column tiles
    const int a_row = lane & 31, k_half = lane >> 5;

    const uint32_t n_chunks = m_end > m_begin ? (m_end - m_begin + FIR_KC - 1) / FIR_KC : 0;
    if (n_chunks) {
        load_chunk(m_begin);
        store_chunk(0);
    }
    __syncthreads();
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) load_chunk(m_begin + (c + 1) * FIR_KC);  // in flight during the MFMAs below
        const float* a = As + buf * 32 * FIR_PITCH + a_row * FIR_PITCH;
        const float* hw = Hw + buf * (256 + FIR_KC) + (FIR_KC - 1) + (lane & 31);
        // operands of step kk+2 are read while the MFMAs of step kk run (the matrix pipe takes 64 cycles each)
        float av_n = a[k_half], b0_n = hw[ct0 * 32 - k_half], b1_n = hw[ct1 * 32 - k_half];
#pragma unroll
        for (int kk = 0; kk < FIR_KC; kk += 2) {  // ascending m: the fmaf chain order of the SPEC
            const float av = av_n, b0 = b0_n, b1 = b1_n;
            if (kk + 2 < FIR_KC) {
                const int k = kk + 2 + k_half;
                av_n = a[k];
                b0_n = hw[ct0 * 32 - k];
                b1_n = hw[ct1 * 32 - k];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads above ahead of the MFMAs below
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
        }
        if (c + 1 < n_chunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // partials[seg][row][col]: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const size_t row_pitch = (size_t)gridDim.z / col_groups * n_pad;  // K * n_pad
    float* P = partials + ((size_t)seg * n_rows_pad + row0) * row_pitch + (size_t)kb * n_pad + ib;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        P[(size_t)rr * row_pitch + ct0 * 32 + (lane & 31)] = acc0[r];
        P[(size_t)rr * row_pitch + ct1 * 32 + (lane & 31)] = acc1[r];
    }
}

// segment partials added in segment order; writes the node outputs, clears their silence flags, advances the ring
__global__ void k_fir_reduce(DevView v, const FirRow* __restrict__ rows, int n_rows, const float* __restrict__ partials,
                             int n_segs, int n_rows_pad, int n_pad) {
    int r = blockIdx.x;
    if (r >= n_rows) return;
    const FirRow row = rows[r];
    if (row.state < 0) return;
    const uint32_t kb = blockIdx.y, K = gridDim.y;
    const size_t row_pitch = (size_t)K * n_pad;
    float* out = v.pool + (size_t)kb * v.pool_blk_stride + (size_t)row.out_buf * v.stride;
    for (int i = threadIdx.x; i < v.frames; i += blockDim.x) {
        float t = partials[(size_t)r * row_pitch + (size_t)kb * n_pad + i];
        for (int sgm = 1; sgm < n_segs; ++sgm) t = t + partials[((size_t)sgm * n_rows_pad + r) * row_pitch + (size_t)kb * n_pad + i];
        // SPEC: "+ (+0.0f)" — a sum that underflowed to -0.0 becomes +0.0 (every other value is unchanged).  Without
        // it the sign of such a zero would depend on whether the window ends in zero padding (a trailing (+0)(+0) term
        // turns -0.0 into +0.0): an implementation detail of the GEMM tiling (found by the DAG fuzz test: the last frame
        // of a block has no trailing out-of-band term in the definition, the tiled GEMM always has padding)
        out[i] = t + 0.0f;
    }
    if (threadIdx.x == 0) {
        v.flags[(size_t)kb * v.flags_blk_stride + row.out_buf] = 0;
        if (row.ch == 0 && kb == 0) {
            NodeState* s = &v.states[row.state];
            s->playhead = (s->playhead + (uint64_t)K * (uint64_t)v.frames) % s->loop_end;
        }
    }
}

