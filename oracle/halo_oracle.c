/* TEST INFRASTRUCTURE ONLY -- placeholder translation unit; the HaloExchange oracle lives in oracle/halo.py
 * (pure index arithmetic, numpy). */
int orc_halo_placeholder(void) { return 0; }
