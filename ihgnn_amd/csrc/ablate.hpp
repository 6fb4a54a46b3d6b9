// ablate.hpp - compile-time switches of the ABLATION builds (tools/ab_variant.sh NAME -DIHG_ABL_<SWITCH> ...; never the product build).
// An ablation build removes one class of work from a kernel - its result is WRONG on purpose - so that the time that class costs inside the
// kernel can be read off (profiles/r4/abl_members.txt).  The product build defines none of the macros: every switch below is a constexpr false and
// the branches it guards fold away; ihg_ablation_build() reports abl::any and the binding refuses such a library (tests/test_abi.py).
#pragma once

namespace abl {

#ifdef IHG_ABL_M_NO_G_STORES      // member gradients: no stores of the [E, 2|3, d] member buffer
constexpr bool m_no_g_stores = true;
#else
constexpr bool m_no_g_stores = false;
#endif
#ifdef IHG_ABL_M_NO_DOUT_STORE    // member gradients, gathering form: the hyperedges' cotangents are not stored
constexpr bool m_no_dout_store = true;
#else
constexpr bool m_no_dout_store = false;
#endif
#ifdef IHG_ABL_M_NO_MEMBER_LOADS  // member gradients: the three member rows of h are not loaded (values made up from the ids)
constexpr bool m_no_member_loads = true;
#else
constexpr bool m_no_member_loads = false;
#endif
#ifdef IHG_ABL_M_NO_DY_LOADS      // member gradients: the cotangent rows (dout, or the three gathered dy rows) are not loaded
constexpr bool m_no_dy_loads = true;
#else
constexpr bool m_no_dy_loads = false;
#endif
#ifdef IHG_ABL_M_NO_MFMA          // member gradients: the matrix waves only keep the barriers
constexpr bool m_no_mfma = true;
#else
constexpr bool m_no_mfma = false;
#endif
#ifdef IHG_ABL_M_NO_SPLIT         // member gradients: the bf16 images are written from the raw bits (no split arithmetic)
constexpr bool m_no_split = true;
#else
constexpr bool m_no_split = false;
#endif
#ifdef IHG_ABL_M_NO_USER_SUMS     // member gradients, user-reduced form: no run sums of the user slot
constexpr bool m_no_user_sums = true;
#else
constexpr bool m_no_user_sums = false;
#endif
#ifdef IHG_ABL_M_NO_PRODUCT_RULE  // member gradients: no product rule (and so no member-buffer stores and no user image)
constexpr bool m_no_product_rule = true;
#else
constexpr bool m_no_product_rule = false;
#endif

#ifdef IHG_ABL_N_NO_MFMA           // node-level contraction: the matrix waves only keep the barriers
constexpr bool n_no_mfma = true;
#else
constexpr bool n_no_mfma = false;
#endif
#ifdef IHG_ABL_N_NO_SPLIT          // node-level contraction: no scaling / split / image writes in the service waves
constexpr bool n_no_split = true;
#else
constexpr bool n_no_split = false;
#endif
#ifdef IHG_ABL_N_NO_FIRST          // node-level contraction: no row loads in the service waves (values made up from the row number)
constexpr bool n_no_first = true;
#else
constexpr bool n_no_first = false;
#endif
#ifdef IHG_ABL_N_NO_SHUFFLE        // node-level contraction: the row's scale from the thread's own values (no cross-lane maximum)
constexpr bool n_no_shuffle = true;
#else
constexpr bool n_no_shuffle = false;
#endif
#ifdef IHG_ABL_N_NO_SERVICE        // node-level contraction: the service waves only keep the barriers (the matrix waves' time on their own)
constexpr bool n_no_service = true;
#else
constexpr bool n_no_service = false;
#endif
#ifdef IHG_ABL_N_NO_FRAGMENTS      // node-level contraction: the matrix waves read no fragments from LDS (one set, read once)
constexpr bool n_no_fragments = true;
#else
constexpr bool n_no_fragments = false;
#endif
#ifdef IHG_ABL_N_NO_RELOAD         // node-level contraction: the weight planes are loaded once per workgroup
constexpr bool n_no_reload = true;
#else
constexpr bool n_no_reload = false;
#endif
#ifdef IHG_ABL_D_NO_DW             // node-level backward: no weight-gradient MFMAs
constexpr bool d_no_dw = true;
#else
constexpr bool d_no_dw = false;
#endif
#ifdef IHG_ABL_D_NO_DX             // node-level backward: no input-gradient MFMAs
constexpr bool d_no_dx = true;
#else
constexpr bool d_no_dx = false;
#endif
#ifdef IHG_ABL_D_NO_SPLIT          // node-level backward: the images are not rewritten (no split of the next tile)
constexpr bool d_no_split = true;
#else
constexpr bool d_no_split = false;
#endif
#ifdef IHG_ABL_D_NO_LOADS          // node-level backward: no row loads (values made up)
constexpr bool d_no_loads = true;
#else
constexpr bool d_no_loads = false;
#endif
#ifdef IHG_ABL_D_NO_STORES         // node-level backward: dx is not stored
constexpr bool d_no_stores = true;
#else
constexpr bool d_no_stores = false;
#endif
#ifdef IHG_ABL_M_G_WINDOW          // member gradients: the member buffer's stores land in a 48 MB window (same instructions, no HBM write traffic) - round 5: same time
constexpr bool m_g_window = true;
#else
constexpr bool m_g_window = false;
#endif
#ifdef IHG_ABL_M_G_PARTS           // member gradients: every column part writes a contiguous stream of its own ([part][e][slot][columns]) - round 5: same time
constexpr bool m_g_parts = true;
#else
constexpr bool m_g_parts = false;
#endif
#ifdef IHG_ABL_M_G_PLAIN           // member gradients: plain instead of non-temporal stores of the member buffer - round 5: same time
constexpr bool m_g_plain = true;
#else
constexpr bool m_g_plain = false;
#endif
#ifdef IHG_ABL_M_UNCOND            // member gradients: every store of the service loop unconditional (rows past the end and the phases without a tile store too: wrong there) and no forced delivery
constexpr bool m_uncond = true;
#else
constexpr bool m_uncond = false;
#endif
#ifdef IHG_ABL_M_TRACE             // member gradients: clock stamps at marks inside the phases of one workgroup (results stay correct; still not the product build)
constexpr bool m_trace = true;
#else
constexpr bool m_trace = false;
#endif
#ifdef IHG_ABL_M_EPI_NOMATH        // member gradients: the product rule stores the contraction values as they are (image reads, no arithmetic).  (Storing values that do not come from the image removes the MFMA loop with it: the compiler drops LDS writes nobody reads - an ablation that reads nothing of the image measures an empty matrix role)
constexpr bool m_epi_nomath = true;
#else
constexpr bool m_epi_nomath = false;
#endif
#ifdef IHG_ABL_D_TRACE             // node-level linear backward: clock stamps inside the phases of one workgroup (results stay correct)
constexpr bool d_trace = true;
#else
constexpr bool d_trace = false;
#endif
constexpr bool any = d_trace || m_epi_nomath || m_trace || m_uncond || m_g_window || m_g_parts || m_g_plain || d_no_dw || d_no_dx || d_no_split || d_no_loads || d_no_stores || n_no_service || n_no_fragments || n_no_reload || n_no_mfma || n_no_split || n_no_first || n_no_shuffle || m_no_g_stores || m_no_dout_store || m_no_member_loads || m_no_dy_loads || m_no_mfma || m_no_split || m_no_user_sums || m_no_product_rule;
}  // namespace abl
