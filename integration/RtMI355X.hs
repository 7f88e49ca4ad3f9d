{-# LANGUAGE ForeignFunctionInterface #-}
{- |
RtMI355X -- the binding a maintainer of jrraymond/ray-tracer adds next to src/RayTracer.hs to
run `render` on MI355X GPUs through librt_mi355x.so (C ABI: include/rt_mi355x.h).

NOT COMPILED HERE: the build image has no GHC (SURVEY.md F2).  It is written against the
reference's own modules (Types, RayTracer.generateGrids) and plain `base` + `vector`; the byte
layouts below are those of the C structs (sizes asserted by tests/test_abi_symbols.py on the
ctypes mirrors: rt_bvh_node 32, rt_sphere 24, rt_triangle 56, rt_material 48, rt_light 48,
rt_camera 104 bytes).  The same entry points are exercised end to end from C
(integration/rt_harness.c) and Python (ray-tracer_amd/__init__.py).

Use: in src/Main.hs replace `let img = render rng w` (Main.hs:44) by `let img = renderGPU rng w`
(or `renderMultiGPU n rng w`), add this module to the cabal file with
`extra-libraries: rt_mi355x` and `extra-lib-dirs:` pointing at ray-tracer_amd/lib.
-}
module RtMI355X (renderGPU, renderGPUDeviceTree, renderMultiGPU, flatten, flattenObjects, Flat(..)) where

import Control.Exception (bracket)
import Control.Monad (when, zipWithM_)
import Data.Bits ((.|.))
import Data.Int (Int32)
import Data.List (elemIndex)
import Data.Word (Word32, Word8)
import Foreign
import Foreign.C.String (CString, peekCString)
import Foreign.C.Types
import System.IO.Unsafe (unsafePerformIO)
import System.Random.Mersenne.Pure64 (PureMT)

import RayTracer (generateGrids)   -- the sample table stays the reference's own (RayTracer.hs:462-467)
import Types

data RtCtx     -- opaque rt_ctx
data RtMulti   -- opaque rt_multi

-- long-running calls are `safe` (an `unsafe` call would stall the RTS for the whole frame)
foreign import ccall safe   "rt_init"               c_rt_init    :: CInt -> Ptr (Ptr RtCtx) -> IO CInt
foreign import ccall safe   "rt_shutdown"           c_rt_shutdown :: Ptr RtCtx -> IO ()
foreign import ccall safe   "rt_upload_scene"       c_rt_upload  :: Ptr RtCtx -> Ptr () -> IO CInt
foreign import ccall safe   "rt_build_scene"        c_rt_build   :: Ptr RtCtx -> Ptr () -> Int32 -> Ptr () -> Int32 -> Ptr () -> Int32 -> Ptr () -> Int32 -> IO CInt
foreign import ccall safe   "rt_set_camera"         c_rt_camera  :: Ptr RtCtx -> Ptr () -> IO CInt
foreign import ccall safe   "rt_set_samples"        c_rt_samples :: Ptr RtCtx -> Ptr CFloat -> Int32 -> Int32 -> IO CInt
foreign import ccall safe   "rt_render"             c_rt_render  :: Ptr RtCtx -> Word32 -> Ptr CFloat -> Ptr Word32 -> Ptr () -> IO CInt
foreign import ccall unsafe "rt_last_error"         c_rt_error   :: Ptr RtCtx -> IO CString
foreign import ccall safe   "rt_multi_init"         c_m_init     :: Int32 -> Ptr Int32 -> Ptr (Ptr RtMulti) -> IO CInt
foreign import ccall safe   "rt_multi_shutdown"     c_m_shutdown :: Ptr RtMulti -> IO ()
foreign import ccall safe   "rt_multi_upload_scene" c_m_upload   :: Ptr RtMulti -> Ptr () -> IO CInt
foreign import ccall safe   "rt_multi_set_camera"   c_m_camera   :: Ptr RtMulti -> Ptr () -> IO CInt
foreign import ccall safe   "rt_multi_set_samples"  c_m_samples  :: Ptr RtMulti -> Ptr CFloat -> Int32 -> Int32 -> IO CInt
foreign import ccall safe   "rt_multi_render"       c_m_render   :: Ptr RtMulti -> Word32 -> Ptr CFloat -> Ptr Word8 -> Ptr () -> IO CInt
foreign import ccall safe   "rt_multi_submit"       c_m_submit   :: Ptr RtMulti -> Word32 -> Ptr CFloat -> Ptr Word8 -> IO CInt
foreign import ccall safe   "rt_multi_wait"         c_m_wait     :: Ptr RtMulti -> Ptr () -> IO CInt
foreign import ccall safe   "rt_multi_set_concurrency" c_m_concurrency :: Ptr RtMulti -> Int32 -> IO CInt
foreign import ccall unsafe "rt_multi_last_error"   c_m_error    :: Ptr RtMulti -> IO CString

-- ---------------------------------------------------------------------------------------------
-- flattening of `World`
-- ---------------------------------------------------------------------------------------------

-- | One flattened tree node: the six `Box` fields in declaration order (Types.hs:88-93), then
-- (left, right): inner node = indices of the children; leaf = (-1 - first, count) into `fRefs`.
data FlatNode = FlatNode !Box !Int32 !Int32

data Flat = Flat
  { fNodes     :: [FlatNode]            -- node 0 is the root; `Empty` = no nodes
  , fRefs      :: [Word32]              -- leaf object lists, in list order; bit 31 = triangle
  , fSpheres   :: [(Vec3, Float, Int32, Int32)]                   -- pos, rad, material, object id
  , fTriangles :: [(Vec3, Vec3, Vec3, Vec3, Int32, Int32)]        -- a, b, c, STORED normal, material, object id
  , fMaterials :: [Material]            -- de-duplicated (every Object embeds its own copy)
  }

-- | Depth-first (pre-order) flattening of `data BVH = Node !BVH !BVH !Box | Leaf ![Object] !Box |
-- Empty` (Types.hs:97-100).  A node's index is the number of nodes emitted before it; the left
-- subtree follows its parent immediately, the right subtree follows the left one.  Objects are
-- numbered in the order the walk meets them (left to right, leaf list order) -- that number is the
-- object id reported in primary_hit_ids; any consistent numbering works, the renderer only hands it
-- back.  An `Empty` child of a `Node` cannot be expressed in rt_bvh_node and does not occur:
-- sahBVH/noBVH never build one (BoundingVolumeHierarchy.hs:34-51,130-131).
flatten :: BVH -> Flat
flatten Empty = Flat [] [] [] [] []
flatten bvh0  = finish (go bvh0 start)
  where
    start = St 0 [] [] [] [] [] 0
    finish st = Flat (reverse (stNodes st)) (reverse (stRefs st)) (reverse (stSph st)) (reverse (stTri st))
                     (reverse (stMats st))

    -- returns the state after emitting the subtree; the subtree's root index is stN before the call
    go :: BVH -> St -> St
    go Empty _ = error "RtMI355X.flatten: Empty below a Node"
    go (Leaf objs box) st =
      let first = fromIntegral (length (stRefs st)) :: Int32
          st1   = st { stN = stN st + 1
                     , stNodes = FlatNode box (-1 - first) (fromIntegral (length objs)) : stNodes st }
      in foldl addObject st1 objs
    go (Node l r box) st =
      let me   = stN st
          -- reserve this node, emit the left subtree (root index me + 1), then the right one
          stL  = go l st { stN = me + 1, stNodes = placeholder : stNodes st }
          rIx  = stN stL
          stR  = go r stL
          node = FlatNode box (fromIntegral (me + 1)) (fromIntegral rIx)
      in stR { stNodes = replaceFromEnd me node (stNodes stR) }
    placeholder = FlatNode EmptyBox 0 0
    -- stNodes is kept reversed: element i (from the front of the final list) sits at
    -- position (length - 1 - i)
    replaceFromEnd i x xs = let k = length xs - 1 - i in take k xs ++ [x] ++ drop (k + 1) xs

-- | One more object: its record, its reference in list order, its material's index in the
-- de-duplicated table.
addObject :: St -> Object -> St
addObject st (Sphere c r m) =
  let (mi, st') = materialIndex m st
      ref = fromIntegral (length (stSph st')) :: Word32
  in st' { stRefs = ref : stRefs st'
         , stSph = (c, r, mi, stObj st') : stSph st', stObj = stObj st' + 1 }
addObject st (Triangle a b c n m) =
  let (mi, st') = materialIndex m st
      ref = 0x80000000 .|. fromIntegral (length (stTri st')) :: Word32
  in st' { stRefs = ref : stRefs st'
         , stTri = (a, b, c, n, mi, stObj st') : stTri st', stObj = stObj st' + 1 }

materialIndex :: Material -> St -> (Int32, St)
materialIndex m st = case elemIndex m (reverse (stMats st)) of
  Just i  -> (fromIntegral i, st)
  Nothing -> (fromIntegral (length (stMats st)), st { stMats = m : stMats st })

-- | The objects alone, in list order and without a tree (`fNodes` empty): the input of rt_build_scene, which
-- builds its own tree on the GPU.
flattenObjects :: [Object] -> Flat
flattenObjects objs =
  let st = foldl addObject (St 0 [] [] [] [] [] 0) objs
  in Flat [] (reverse (stRefs st)) (reverse (stSph st)) (reverse (stTri st)) (reverse (stMats st))

data St = St { stN :: !Int, stNodes :: [FlatNode], stRefs :: [Word32]
             , stSph :: [(Vec3, Float, Int32, Int32)], stTri :: [(Vec3, Vec3, Vec3, Vec3, Int32, Int32)]
             , stMats :: [Material], stObj :: !Int32 }

-- ---------------------------------------------------------------------------------------------
-- marshalling (explicit byte offsets of the C structs in include/rt_mi355x.h)
-- ---------------------------------------------------------------------------------------------

pokeF :: Ptr a -> Int -> Float -> IO ()
pokeF p off x = pokeByteOff p off (realToFrac x :: CFloat)

pokeV3 :: Ptr a -> Int -> Vec3 -> IO ()
pokeV3 p off (Vec3 x y z) = pokeF p off x >> pokeF p (off + 4) y >> pokeF p (off + 8) z

pokeC3 :: Ptr a -> Int -> Color -> IO ()
pokeC3 p off (Color r g b) = pokeF p off r >> pokeF p (off + 4) g >> pokeF p (off + 8) b

pokeI :: Ptr a -> Int -> Int32 -> IO ()
pokeI = pokeByteOff

-- | rt_scene: six (pointer, int32 count) pairs, each pair padded to 16 bytes on LP64.
withScene :: Flat -> [Light] -> (Ptr () -> IO a) -> IO a
withScene fl lights k =
  withRecords 32 (fNodes fl)     pokeNode     $ \pn ->
  withArrayLen' (fRefs fl)                    $ \pr ->
  withRecords 24 (fSpheres fl)   pokeSphere   $ \ps ->
  withRecords 56 (fTriangles fl) pokeTriangle $ \pt ->
  withRecords 48 (fMaterials fl) pokeMaterial $ \pm ->
  withRecords 48 lights          pokeLight    $ \pl ->
  allocaBytes 96 $ \sc -> do
    let field i ptr n = pokeByteOff sc (16 * i) ptr >> pokeI sc (16 * i + 8) (fromIntegral n)
    field 0 pn (length (fNodes fl))
    field 1 (castPtr pr) (length (fRefs fl))
    field 2 ps (length (fSpheres fl))
    field 3 pt (length (fTriangles fl))
    field 4 pm (length (fMaterials fl))
    field 5 pl (length lights)
    k sc
  where
    withArrayLen' xs f = withArray xs f
    pokeNode p (FlatNode box l r) = do
      case box of
        Box x0 x1 y0 y1 z0 z1 -> zipWithM_ (\i v -> pokeF p (4 * i) v) [0 ..] [x0, x1, y0, y1, z0, z1]
        EmptyBox -> error "RtMI355X: EmptyBox in a built BVH"   -- hitsBox errors on it too (BVH.hs:192)
      pokeI p 24 l >> pokeI p 28 r

-- | A table of fixed-size C records for the duration of a call.
-- (C heap, not allocaBytes: a mesh's tables are megabytes, which do not belong in pinned
--  byte arrays on the Haskell heap; the library copies them during the call and keeps no pointer)
withRecords :: Int -> [r] -> (Ptr () -> r -> IO ()) -> (Ptr () -> IO b) -> IO b
withRecords size rs pk f = bracket (mallocBytes (max 1 (size * length rs))) free $ \p -> do
  zipWithM_ (\i r -> pk (p `plusPtr` (size * i)) r) [0 ..] rs
  f p

pokeSphere :: Ptr () -> (Vec3, Float, Int32, Int32) -> IO ()
pokeSphere p (c, r, mi, oid) = pokeV3 p 0 c >> pokeF p 12 r >> pokeI p 16 mi >> pokeI p 20 oid

pokeTriangle :: Ptr () -> (Vec3, Vec3, Vec3, Vec3, Int32, Int32) -> IO ()
pokeTriangle p (a, b, c, n, mi, oid) =
  pokeV3 p 0 a >> pokeV3 p 12 b >> pokeV3 p 24 c >> pokeV3 p 36 n >> pokeI p 48 mi >> pokeI p 52 oid

-- matAtten already holds the LOG of the attenuation colour (makeMaterial, Surfaces.hs:22-24)
pokeMaterial :: Ptr () -> Material -> IO ()
pokeMaterial p (Material d s ph rl ri at) =
  pokeC3 p 0 d >> pokeC3 p 12 s >> pokeF p 24 ph >> pokeF p 28 rl >> pokeF p 32 ri >> pokeC3 p 36 at

pokeLight :: Ptr () -> Light -> IO ()
pokeLight p (Light c a b col) = pokeV3 p 0 c >> pokeV3 p 12 a >> pokeV3 p 24 b >> pokeC3 p 36 col

-- | rt_camera (104 bytes): the scalar fields of `World` (Types.hs:132-146).
withCamera :: World -> (Ptr () -> IO a) -> IO a
withCamera w k = allocaBytes 104 $ \p -> do
  pokeI p 0 (round (wImgWd w)) >> pokeI p 4 (round (wImgHt w))
  pokeF p 8 (wViewWd w) >> pokeF p 12 (wViewHt w) >> pokeF p 16 (wViewDt w)
  pokeI p 20 (fromIntegral (wAntiAliasing w)) >> pokeF p 24 (wLens w)
  pokeV3 p 28 (wUp w) >> pokeV3 p 40 (wEye w)
  let (u, v, ww) = wCamera w
  pokeV3 p 52 u >> pokeV3 p 64 v >> pokeV3 p 76 ww
  pokeC3 p 88 (wAmbient w)
  pokeI p 100 (fromIntegral (wMaxDepth w))
  k p

-- | The `generateGrids` table before `cycle` (RayTracer.hs:20, 462-467): (W+10) grids of aa*aa F6.
sampleTable :: PureMT -> World -> ([CFloat], Int, Int)
sampleTable rng w =
  let wd = round (wImgWd w) :: Int
      aa = wAntiAliasing w
      grids = take (wd + 10) (generateGrids rng (wd + 10) aa)
  in ([realToFrac x | g <- grids, F6 a b c d e f <- g, x <- [a, b, c, d, e, f]], wd + 10, aa * aa)

check :: String -> IO String -> CInt -> IO ()
check what why rc = when (rc /= 0) $ do
  msg <- why
  ioError (userError (what ++ " failed (" ++ show rc ++ "): " ++ msg))

toColors :: Int -> Ptr CFloat -> IO [Color]
toColors n p = do
  xs <- peekArray (3 * n) p
  let go (r : g : b : rest) = Color (realToFrac r) (realToFrac g) (realToFrac b) : go rest
      go _ = []
  return (go xs)

-- | Drop-in for `render` (RayTracer.hs:18): same type, same list order (top row first).
renderGPU :: PureMT -> World -> [Color]
renderGPU rng w = unsafePerformIO $
  alloca $ \pctx -> do
    check "rt_init" (return "no GPU?") =<< c_rt_init 0 pctx
    bracket (peek pctx) c_rt_shutdown $ \ctx -> do
      let err = c_rt_error ctx >>= peekCString
          (f6, nGrids, spp) = sampleTable rng w
          n = round (wImgWd w) * round (wImgHt w)
      withScene (flatten (wObjects w)) (wLights w) $ \sc -> check "rt_upload_scene" err =<< c_rt_upload ctx sc
      withCamera w $ \pc -> check "rt_set_camera" err =<< c_rt_camera ctx pc
      withArray f6 $ \p -> check "rt_set_samples" err =<< c_rt_samples ctx p (fromIntegral nGrids) (fromIntegral spp)
      allocaArray (3 * n) $ \out -> do
        check "rt_render" err =<< c_rt_render ctx 0 out nullPtr nullPtr
        toColors n out
{-# NOINLINE renderGPU #-}

-- | `render` with the tree built ON THE GPU (rt_build_scene: a linear BVH, one object per leaf) from the
-- objects alone -- for a frame of an animation whose objects moved too far for last frame's tree and for which
-- `sahBVH` on the host would cost more than the frame (100 000 triangles: 0.2 s against 2 ms).  `wObjects` of
-- the world is not looked at.  A TREE-CHANGING mode: hits that tie in distance may be won by another object than
-- under the `sahBVH` tree (include/rt_mi355x.h).
renderGPUDeviceTree :: PureMT -> World -> [Object] -> [Color]
renderGPUDeviceTree rng w objs = unsafePerformIO $
  alloca $ \pctx -> do
    check "rt_init" (return "no GPU?") =<< c_rt_init 0 pctx
    bracket (peek pctx) c_rt_shutdown $ \ctx -> do
      let err = c_rt_error ctx >>= peekCString
          (f6, nGrids, spp) = sampleTable rng w
          n = round (wImgWd w) * round (wImgHt w)
          fl = flattenObjects objs
          len xs = fromIntegral (length xs)   -- (a function binding: used at four list types)
      withRecords 24 (fSpheres fl)   pokeSphere   $ \ps ->
        withRecords 56 (fTriangles fl) pokeTriangle $ \pt ->
        withRecords 48 (fMaterials fl) pokeMaterial $ \pm ->
        withRecords 48 (wLights w)     pokeLight    $ \pl ->
          check "rt_build_scene" err =<< c_rt_build ctx ps (len (fSpheres fl)) pt (len (fTriangles fl))
                                                     pm (len (fMaterials fl)) pl (len (wLights w))
      withCamera w $ \pc -> check "rt_set_camera" err =<< c_rt_camera ctx pc
      withArray f6 $ \p -> check "rt_set_samples" err =<< c_rt_samples ctx p (fromIntegral nGrids) (fromIntegral spp)
      allocaArray (3 * n) $ \out -> do
        check "rt_render" err =<< c_rt_render ctx 0 out nullPtr nullPtr
        toColors n out
{-# NOINLINE renderGPUDeviceTree #-}

-- | The same frame rendered on the first n GPUs of the node (tiles t mod n, RCCL gather on GPU 0).
renderMultiGPU :: Int -> PureMT -> World -> [Color]
renderMultiGPU nGpus rng w = unsafePerformIO $
  alloca $ \pm -> do
    poke pm nullPtr
    -- rt_multi_init hands back a handle even when it fails (so that the error text can be read, and
    -- because some ranks' contexts and streams may exist already): the bracket is entered BEFORE the
    -- return code is looked at, so that a failed init is shut down like a successful one
    bracket (do rc <- c_m_init (fromIntegral nGpus) nullPtr pm
                m0 <- peek pm
                return (rc, m0))
            (\(_, m0) -> when (m0 /= nullPtr) (c_m_shutdown m0)) $ \(rc, m) -> do
      check "rt_multi_init" (if m == nullPtr then return "" else c_m_error m >>= peekCString) rc
      let err = c_m_error m >>= peekCString
          (f6, nGrids, spp) = sampleTable rng w
          n = round (wImgWd w) * round (wImgHt w)
      withScene (flatten (wObjects w)) (wLights w) $ \sc -> check "rt_multi_upload_scene" err =<< c_m_upload m sc
      withCamera w $ \pc -> check "rt_multi_set_camera" err =<< c_m_camera m pc
      withArray f6 $ \p -> check "rt_multi_set_samples" err =<< c_m_samples m p (fromIntegral nGrids) (fromIntegral spp)
      allocaArray (3 * n) $ \out -> do
        check "rt_multi_render" err =<< c_m_render m 0 out nullPtr nullPtr
        toColors n out
{-# NOINLINE renderMultiGPU #-}
